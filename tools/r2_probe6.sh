#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -12
for MP in 1 0; do
LCX_MERGED_PASS=$MP python bench.py --workload c3 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('c3 merged=$MP', d['value'], d['ms_per_step'], r['frac'], r['kernel'], d['config']['x_passes_per_iteration'], {k:(round(v['avg_us']),round(v['TFLOPs'],1)) for k,v in r['use_sites'].items()}, r['iteration'])"
done
LCX_MERGED_PASS=1 python bench.py --workload c2f32 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('c2f32 merged=1', d['value'], d['ms_per_step'], r['frac'], r['kernel'], d['config']['x_passes_per_iteration'])"
LCX_MERGED_PASS=0 python bench.py --workload c2f32 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('c2f32 merged=0', d['value'], d['ms_per_step'], r['frac'], r['kernel'], d['config']['x_passes_per_iteration'])"
