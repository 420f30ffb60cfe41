#!/bin/bash
# round-2 third GPU session: suite with lcx_iterate as the default path, default bench line, c2 kernel trace (gaps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_bench_default.err | tail -1 > gpurun_out/r02_bench_default.json
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r02_bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_default.json')); c=d['config']['c2']
print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['x_passes_per_iteration'])
print('c2', c['value'], c['ms_per_step'], c['roofline']['frac'], c['x_passes_per_iteration'], c['windows']['ms_per_step_walk_min_median_max'], c['roofline']['iteration'])
"
LCX_HOST_LOOP=1 python bench.py --workload c2 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 host loop', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python3 $R/bench.py --workload c2 --no-extras --repeats 2 > $R/gpurun_out/prof_c2.log 2>&1
TRACE=$(find $R/gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_gaps.py "$TRACE" | tee $R/gpurun_out/r02_trace_gaps_c2.txt | head -30
