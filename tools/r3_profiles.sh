#!/bin/bash
# round-3 profile session: rocprofv3 kernel stats + PMC passes of the bench command for c3 / c2 / c4shard (stamped with the library
# source hash), then the driver's bench command with those profiles in place
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python3 __graft_entry__.py || exit 1
for WL in c3 c2 c4shard; do
  bash tools/gpu_prof.sh r03 $WL 2>&1 | tail -30
  cp gpurun_out/pmc_traffic_$WL.json profiles/pmc_traffic_$WL.json
  cp gpurun_out/r03_rocprof_kernel_stats_$WL.csv gpurun_out/r03_rocprof_kernel_stats_$WL.meta.json profiles/
  cp gpurun_out/r03_trace_gaps_$WL.txt profiles/ 2>/dev/null
done
cd $R
# the profiles must travel back: profiles/ is not merged by gpurun, gpurun_out/ is
mkdir -p gpurun_out/r3_profiles && cp profiles/pmc_traffic_*.json profiles/r03_rocprof_kernel_stats_* profiles/r03_trace_gaps_* gpurun_out/r3_profiles/
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r3_profiles/bench_default.err | tail -1 > gpurun_out/r3_profiles/bench_default.json
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r3_profiles/bench_default.err
python -c "
import json
d=json.load(open('gpurun_out/r3_profiles/bench_default.json')); c=d['config']['c2']; r=d['roofline']
print('c3', d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'], r['mfma_util_pmc'], d['cpu_baseline']['value'])
r=c['roofline']; print('c2', c['value'], c['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'], c['cpu_baseline']['value'])
c4=d['config']['c4shard']; r=c4['roofline']; print('c4shard', c4['value'], c4['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'])
"
