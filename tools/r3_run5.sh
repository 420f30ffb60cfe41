#!/bin/bash
mkdir -p gpurun_out/r3_run5
for flag in 0 1; do
  for wl in c2 c2f32 tiny; do
  LCX_SIDE_STREAM=$flag python bench.py --workload $wl --no-extras --steps 30 --warmup 5 2>gpurun_out/r3_run5/${wl}_side_$flag.err | tail -1 > gpurun_out/r3_run5/${wl}_side_$flag.json
  python -c "
import json; d=json.load(open('gpurun_out/r3_run5/${wl}_side_$flag.json')); print('$wl LCX_SIDE_STREAM=$flag', round(d['value'],1), round(d['ms_per_step'],4), d['config']['x_passes_per_iteration'], round(d['roofline']['frac'],3), d['config']['windows']['ms_per_step_walk_min_median_max'])"
  done
done
python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -5
