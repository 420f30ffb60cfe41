#!/bin/bash
# round-2 closing GPU session: the suite, smoke, rocprofv3 kernel stats + PMC passes of the bench command for c3 / c2 / c4shard
# (stamped with the library source hash), then the driver's bench command with those profiles in place
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for WL in c3 c2 c4shard; do
  bash tools/gpu_prof.sh r02 $WL 2>&1 | tail -40
  cp gpurun_out/pmc_traffic_$WL.json profiles/pmc_traffic_$WL.json
  cp gpurun_out/r02_rocprof_kernel_stats_$WL.csv gpurun_out/r02_rocprof_kernel_stats_$WL.meta.json profiles/
done
cd $R
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_bench_default.err | tail -1 > gpurun_out/r02_bench_default.json
echo "bench default: $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r02_bench_default.err
python bench.py --workload c4shard --no-extras 2>/dev/null | tail -1 > gpurun_out/r02_bench_c4shard.json
# the figures must not move with the driver's step count: the same lines at --steps 5 and --steps 70
{
  for SW in "5 1" "20 5" "70 7"; do
    set -- $SW
    python bench.py --workload c3 --no-extras --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['config']['windows']
print('c3 --steps %3d --warmup %d: %.2f it/s, %.3f ms per step (%.2f X passes, %.2f trials per iteration), frac %.3f, %d timed iterations in %.2f s' % (d['steps'], d['warmup'], d['value'], d['ms_per_step'], d['config']['x_passes_per_iteration'], d['config']['line_search_trials_per_iteration'], d['roofline']['frac'], w['timed_iterations'], w['timed_seconds']))"
    python bench.py --workload c2 --no-extras --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['config']['windows']
print('c2 --steps %3d --warmup %d: %.1f it/s, %.4f ms per step, frac %.3f, %d timed iterations in %.2f s, %d walks' % (d['steps'], d['warmup'], d['value'], d['ms_per_step'], d['roofline']['frac'], w['timed_iterations'], w['timed_seconds'], w['walks_timed']))"
  done
} | tee gpurun_out/r02_step_count_independence.txt
python -c "
import json
d=json.load(open('gpurun_out/r02_bench_default.json')); c=d['config']['c2']; r=d['roofline']
print('c3', d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'], r['mfma_util_pmc'], d['cpu_baseline']['value'])
r=c['roofline']; print('c2', c['value'], c['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'], c['cpu_baseline']['value'], c['get_covariance']['seconds_each_call'], d['config']['get_covariance_c5_standin']['seconds_each_call'])
d=json.load(open('gpurun_out/r02_bench_c4shard.json')); r=d['roofline']
print('c4shard', d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['rocprofv3_avg_kernel_us'], r['avg_launch_us'])
"
