set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 70 --warmup 7 2>&1 | tail -3 | tee gpurun_out/bench_c2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_c2 | head -20
