#!/bin/bash
mkdir -p gpurun_out/r3_run4
for flag in 0 1; do
  LCX_MERGED_PASS=$flag python bench.py --workload c2 --no-extras --steps 30 --warmup 5 2>gpurun_out/r3_run4/c2_merged_$flag.err | tail -1 > gpurun_out/r3_run4/c2_merged_$flag.json
  python -c "
import json; d=json.load(open('gpurun_out/r3_run4/c2_merged_$flag.json')); print('LCX_MERGED_PASS=$flag', d['value'], d['ms_per_step'], d['config']['x_passes_per_iteration'], d['roofline']['frac'], {k: (v['launches'], round(v['avg_us'],1)) for k,v in d['roofline']['use_sites'].items()})"
done
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "library_loop or merged or big5 or planted_small or config2_steps or step_level" 2>&1 | tail -5
