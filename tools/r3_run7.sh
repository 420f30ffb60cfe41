#!/bin/bash
mkdir -p gpurun_out/r3_run7
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "step_level or more_than" > gpurun_out/r3_run7/wide.log 2>&1
echo "wide tests rc=$?"; tail -25 gpurun_out/r3_run7/wide.log
