#!/bin/bash
# round-2 second GPU session: the suite on the reworked engine (MFMA covariance, staging pipeline), the tile-shape probe of
# the float32 large-shard pass, and the new default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 600 ./tools/gemm_probe8 odd > gpurun_out/r02_gemm_probe8.txt 2>&1
timeout 900 ./tools/gemm_probe8 c3 >> gpurun_out/r02_gemm_probe8.txt 2>&1
timeout 600 ./tools/gemm_probe8 c4 >> gpurun_out/r02_gemm_probe8.txt 2>&1
cat gpurun_out/r02_gemm_probe8.txt
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_bench_default.err | tail -1 > gpurun_out/r02_bench_default.json
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -5 gpurun_out/r02_bench_default.err
cut -c1-3000 gpurun_out/r02_bench_default.json
