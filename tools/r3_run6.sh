#!/bin/bash
mkdir -p gpurun_out/r3_run6
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "single_copy" > gpurun_out/r3_run6/single_copy.log 2>&1
echo "single_copy tests rc=$?"; tail -5 gpurun_out/r3_run6/single_copy.log
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "unsharded" > gpurun_out/r3_run6/c4full_test.log 2>&1
echo "c4full test rc=$?"; tail -8 gpurun_out/r3_run6/c4full_test.log
for flag in 0 1; do
  for wl in c3 c4shard; do
  LCX_SINGLE_COPY=$flag python bench.py --workload $wl --no-extras --steps 10 --warmup 3 --repeats 1 2>gpurun_out/r3_run6/${wl}_single_$flag.err | tail -1 > gpurun_out/r3_run6/${wl}_single_$flag.json
  python -c "
import json; d=json.load(open('gpurun_out/r3_run6/${wl}_single_$flag.json')); print('$wl LCX_SINGLE_COPY=$flag', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['roofline']['kernel'], {k: round(v['avg_us'],1) for k,v in d['roofline']['use_sites'].items()}, d['config']['bytes_resident'])"
  done
done
python bench.py --workload c4full --no-extras --steps 3 --warmup 1 --repeats 1 2>gpurun_out/r3_run6/c4full.err | tail -1 > gpurun_out/r3_run6/c4full.json
python -c "
import json; d=json.load(open('gpurun_out/r3_run6/c4full.json')); print('c4full', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), d['roofline']['kernel'], {k: round(v['avg_us'],1) for k,v in d['roofline']['use_sites'].items()}, d['config']['bytes_resident'])"
tail -3 gpurun_out/r3_run6/c4full.err
