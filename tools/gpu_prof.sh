#!/bin/bash
# rocprofv3 kernel stats + PMC HBM traffic of the bench command for one workload (GPU box).  The library is built FIRST,
# as its own step: nothing may compile inside a profiled, GPU-initialised process.
#   bash tools/gpu_prof.sh <tag> <workload>[_split] [extra bench args]
# <workload>_split: the same workload with the float32 X passes on the bf16 matrix pipe (bench --f32-gemm split); outputs named so
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; NAME=$2; shift 2
WL=${NAME%_split}
ARGS="--workload $WL --no-extras $*"
[ "$NAME" != "$WL" ] && ARGS="$ARGS --f32-gemm split"
WL=$NAME
mkdir -p $R/gpurun_out
cd $R && python3 __graft_entry__.py || exit 1
HASH=$(python3 -c "import __graft_entry__ as g; print(g._src_hash())")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -o $WL -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$WL.log 2>&1
tail -1 $R/gpurun_out/prof_$WL.log | cut -c1-200
STATS=$(find $R/gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1)
cp "$STATS" $R/gpurun_out/${TAG}_rocprof_kernel_stats_$WL.csv && head -8 $R/gpurun_out/${TAG}_rocprof_kernel_stats_$WL.csv | cut -c1-220
echo "{\"lib_src_hash\": \"$HASH\", \"command\": \"rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS\"}" > $R/gpurun_out/${TAG}_rocprof_kernel_stats_$WL.meta.json
TRACE=$(find $R/gpurun_out/prof_$WL -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_gaps.py "$TRACE" > $R/gpurun_out/${TAG}_trace_gaps_$WL.txt 2>&1; head -24 $R/gpurun_out/${TAG}_trace_gaps_$WL.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${C}_$WL -o $WL -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${C}_$WL.log 2>&1
  tail -1 $R/gpurun_out/pmc_${C}_$WL.log | cut -c1-120
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/gpurun_out/pmc_MFMA_$WL -o $WL -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_MFMA_$WL.log 2>&1
tail -1 $R/gpurun_out/pmc_MFMA_$WL.log | cut -c1-120
python3 $R/tools/pmc_traffic.py --workload $WL --fetch $R/gpurun_out/pmc_FETCH_SIZE_$WL --write $R/gpurun_out/pmc_WRITE_SIZE_$WL --mfma $R/gpurun_out/pmc_MFMA_$WL \
  --lib-src-hash $HASH --command "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py $ARGS" --out $R/gpurun_out/pmc_traffic_$WL.json
find $R/gpurun_out -name "*counter_collection.csv" -size +2M -delete
find $R/gpurun_out -name "*kernel_trace.csv" -size +8M -delete
