#!/bin/bash
# round-2 closing GPU session: the suite, smoke, rocprofv3 kernel stats + PMC passes of the bench command for c3 / c2 / c4shard
# (stamped with the library source hash), then the driver's bench command with those profiles in place
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for WL in c3 c2 c4shard; do
  bash tools/gpu_prof.sh r02 $WL 2>&1 | tail -40
  cp gpurun_out/pmc_traffic_$WL.json profiles/pmc_traffic_$WL.json
  cp gpurun_out/r02_rocprof_kernel_stats_$WL.csv gpurun_out/r02_rocprof_kernel_stats_$WL.meta.json profiles/
done
cd $R
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_bench_default.err | tail -1 > gpurun_out/r02_bench_default.json
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r02_bench_default.err
python bench.py --workload c4shard --no-extras 2>/dev/null | tail -1 > gpurun_out/r02_bench_c4shard.json
python -c "
import json
d=json.load(open('gpurun_out/r02_bench_default.json')); c=d['config']['c2']; r=d['roofline']
print('c3', d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'], r['mfma_util_pmc'], d['cpu_baseline']['value'])
r=c['roofline']; print('c2', c['value'], c['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'], c['cpu_baseline']['value'], c['get_covariance']['seconds_each_call'], d['config']['get_covariance_c5_standin']['seconds_each_call'])
d=json.load(open('gpurun_out/r02_bench_c4shard.json')); r=d['roofline']
print('c4shard', d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'])
"
