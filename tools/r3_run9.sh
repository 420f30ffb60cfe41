#!/bin/bash
mkdir -p gpurun_out/r3_run9
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -s -k "later_trials" 2>&1 | tail -12
for wl in c3 c4shard c2; do
  for ls in exact exact-y; do
    python bench.py --workload $wl --no-extras --steps 20 --warmup 5 --repeats 1 --line-search $ls 2>gpurun_out/r3_run9/${wl}_$ls.err | tail -1 > gpurun_out/r3_run9/${wl}_$ls.json
    python -c "
import json; d=json.load(open('gpurun_out/r3_run9/${wl}_$ls.json')); print('$wl $ls', round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms; X passes', round(d['config']['x_passes_per_iteration'],3), 'trials', round(d['config']['line_search_trials_per_iteration'],3), 'final TC', d['config']['final_TC'])"
  done
done
