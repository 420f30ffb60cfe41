python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 2>&1 | tail -1 | tee gpurun_out/bench_c2_c.json
