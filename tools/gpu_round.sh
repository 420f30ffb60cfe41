#!/bin/bash
# One GPU-box session: parity tests, smoke, the bench line, rocprofv3 kernel stats and PMC traffic of
# the same command.  Run through gpurun from the repo root; everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
WL=${2:-c2}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --workload $WL 2>gpurun_out/bench_$WL.err | tail -1 > gpurun_out/${TAG}_bench_$WL.json
cut -c1-900 gpurun_out/${TAG}_bench_$WL.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -o $WL -- python3 $R/bench.py --workload $WL --cpu-iters-per-stage 0 --no-also-linear > $R/gpurun_out/prof_$WL.log 2>&1
tail -1 $R/gpurun_out/prof_$WL.log | cut -c1-300
STATS=$(find $R/gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1)
cp "$STATS" $R/gpurun_out/${TAG}_rocprof_kernel_stats_$WL.csv && head -12 $R/gpurun_out/${TAG}_rocprof_kernel_stats_$WL.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${C}_$WL -o $WL -- python3 $R/bench.py --workload $WL --steps 14 --warmup 7 --cpu-iters-per-stage 0 --no-also-linear > $R/gpurun_out/pmc_${C}_$WL.log 2>&1
  tail -1 $R/gpurun_out/pmc_${C}_$WL.log | cut -c1-200
done
python3 $R/tools/pmc_traffic.py --workload $WL --fetch $R/gpurun_out/pmc_FETCH_SIZE_$WL --write $R/gpurun_out/pmc_WRITE_SIZE_$WL \
  --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --workload $WL --steps 14 --warmup 7 --cpu-iters-per-stage 0 --no-also-linear" \
  --out $R/gpurun_out/pmc_traffic_$WL.json
# the raw counter CSVs are large; keep only the summary
find $R/gpurun_out -name "*counter_collection.csv" -size +2M -delete
find $R/gpurun_out -name "*kernel_trace.csv" -size +8M -delete
