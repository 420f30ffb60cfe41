#!/bin/bash
# large-shard session: kernel tests, then the c3 (and optionally c4shard) bench lines + rocprof stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gemm_kernels_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -5
for WL in "$@"; do
  python bench.py --workload $WL --steps 21 --warmup 7 --cpu-iters-per-stage 0 --no-also-linear 2>gpurun_out/bench_$WL.err | tail -1 > gpurun_out/r01_bench_$WL.json
  cut -c1-1800 gpurun_out/r01_bench_$WL.json; tail -3 gpurun_out/bench_$WL.err
done
