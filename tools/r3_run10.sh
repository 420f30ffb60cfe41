#!/bin/bash
mkdir -p gpurun_out/r3_run10
python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -s -k "step_vs_oracle" 2>&1 | tail -8
python -m pytest tests/test_bench_gpu.py -m gpu -x -q -k "default_line" 2>&1 | tail -5
python -m pytest tests/test_distributed_gpu.py -m gpu -x -q -k "float32_large or rccl" 2>&1 | tail -3
