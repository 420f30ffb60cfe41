#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -25
python bench.py --workload 20000x30000x256:f32 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('m=256 f32 20000x30000', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['config']['x_passes_per_iteration'])"
python bench.py --workload 10000x5000x200:f64 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('m=200 f64 10000x5000', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['config']['x_passes_per_iteration'])"
