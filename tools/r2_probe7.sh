#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "merged or library_loop or large_shard" 2>&1 | tail -6
timeout 600 ./tools/gemm_probe8 c3hint > gpurun_out/r02_gemm_probe8_hints.txt 2>&1; cat gpurun_out/r02_gemm_probe8_hints.txt | cut -c1-220
