#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
for i in 1 2 3; do
python bench.py --workload c2 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('c2: %.1f it/s, %.4f ms per step, frac %.3f, pass %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us']))"
done
python tools/small_fit_timing.py 2>/dev/null | tail -4
