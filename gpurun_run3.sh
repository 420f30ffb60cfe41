python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 2>&1 | tail -1 | tee gpurun_out/bench_c2_linear.json | cut -c1-1500
python bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 --line-search exact 2>&1 | tail -1 | tee gpurun_out/bench_c2_exact.json | cut -c1-400
