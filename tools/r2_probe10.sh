#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 ./tools/gemm_probe8 c3hint > gpurun_out/r02_gemm_probe8_hints.txt 2>&1; cat gpurun_out/r02_gemm_probe8_hints.txt | cut -c1-230
bash tools/r2_pmc_probe.sh 2>&1 | grep -A2 "gemm_ct" | cut -c1-330
