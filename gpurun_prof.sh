cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python3 $R/bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 > $R/gpurun_out/prof_c2.log 2>&1
tail -1 $R/gpurun_out/prof_c2.log | cut -c1-200
