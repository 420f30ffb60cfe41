#!/bin/bash
# single-workload bench lines kept beside the driver-shaped one (profiles/r02_bench_{c2,c3}.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for WL in c2 c3; do
  python bench.py --workload $WL --no-extras 2>/dev/null | tail -1 > gpurun_out/r02_bench_$WL.json
  python -c "
import json; d=json.load(open('gpurun_out/r02_bench_$WL.json')); r=d['roofline']
print('$WL', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['rocprofv3_avg_kernel_us'], r['traffic'], d['config']['x_passes_per_iteration'])"
done
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02_bench_default_b.json
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_default_b.json')); c=d['config']['c2']; r=d['roofline']
print('default c3', d['value'], d['ms_per_step'], r['frac'], ' c2', c['value'], c['ms_per_step'], c['roofline']['frac'])"
