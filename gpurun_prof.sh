set -x
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_c2
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log
