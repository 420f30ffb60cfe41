#!/bin/bash
mkdir -p gpurun_out/r3_run8
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "step_level and (case19 or case20 or case21 or case22) or more_than_256" --durations=5 2>&1 | tail -12
for wl in 10000x5000x300:f64 20000x20000x512:f32 50000x100000x300:f32; do
  python bench.py --workload $wl --no-extras --steps 5 --warmup 2 --repeats 1 2>gpurun_out/r3_run8/wide.err | tail -1 > gpurun_out/r3_run8/wide.json
  python -c "
import json; d=json.load(open('gpurun_out/r3_run8/wide.json')); print('$wl', round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms/step', d['roofline']['kernel'], 'frac', round(d['roofline']['frac'],3), {k: (round(v['avg_us'],1), round(v['TFLOPs'],1)) for k,v in d['roofline']['use_sites'].items()}, 'x_passes', round(d['config']['x_passes_per_iteration'],2))"
done
