#!/bin/bash
# round 3, run 1: the new full-size oracle checks, the sharded float32 test, the planted generator check, the new bench line
mkdir -p gpurun_out/r3_run1
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "step_vs_oracle" > gpurun_out/r3_run1/full_size.log 2>&1
echo "full_size rc=$?" >> gpurun_out/r3_run1/summary.txt
python -m pytest tests/test_distributed_gpu.py -m gpu -x -q -k "float32_large" > gpurun_out/r3_run1/sharded_f32.log 2>&1
echo "sharded_f32 rc=$?" >> gpurun_out/r3_run1/summary.txt
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "generated_data" > gpurun_out/r3_run1/gen.log 2>&1
echo "gen rc=$?" >> gpurun_out/r3_run1/summary.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3_run1/bench_default.json 2> gpurun_out/r3_run1/bench_default.err
echo "bench rc=$?" >> gpurun_out/r3_run1/summary.txt
tail -3 gpurun_out/r3_run1/bench_default.err >> gpurun_out/r3_run1/summary.txt
cat gpurun_out/r3_run1/summary.txt
