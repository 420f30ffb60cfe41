#!/bin/bash
# round 3, run 2: predict / invert on the device, and the suites the staging / set_moment changes touch
mkdir -p gpurun_out/r3_run2
python -m pytest tests/test_predict_gpu.py -m gpu -x -q > gpurun_out/r3_run2/predict.log 2>&1
echo "predict rc=$?" >> gpurun_out/r3_run2/summary.txt
python -m pytest tests/test_parity_gpu.py tests/test_syn_gpu.py tests/test_preprocess_gpu.py tests/test_cli.py tests/test_edge_shapes_gpu.py -m gpu -x -q > gpurun_out/r3_run2/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/r3_run2/summary.txt
tail -15 gpurun_out/r3_run2/predict.log
tail -5 gpurun_out/r3_run2/parity.log
cat gpurun_out/r3_run2/summary.txt
