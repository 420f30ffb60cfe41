cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python3 $R/bench.py --steps 70 --warmup 7 --cpu-iters-per-stage 0 > $R/gpurun_out/prof_c2.log 2>&1
tail -1 $R/gpurun_out/prof_c2.log | cut -c1-300
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o c2 -- python3 $R/bench.py --steps 14 --warmup 7 --cpu-iters-per-stage 0 --no-kernel-timing > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o c2 -- python3 $R/bench.py --steps 14 --warmup 7 --cpu-iters-per-stage 0 --no-kernel-timing > $R/gpurun_out/pmc_write.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
