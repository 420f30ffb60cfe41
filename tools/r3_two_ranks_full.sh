#!/bin/bash
# the driver's multi-GPU command at full size, rehearsed on ONE GPU: two ranks share GPU 0, gloo instead of RCCL (RCCL refuses
# two ranks per device) - exercises the default --gpus N path (config-4 shards, same-shard reference, c2_weak block) end to end
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
T0=$(date +%s)
LCX_BENCH_DEVICE=0 LCX_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 2>gpurun_out/r03_two_ranks_full.err | tail -1 > gpurun_out/r03_two_ranks_full.json
echo "rc=$? wall $(( $(date +%s) - T0 )) s"; tail -5 gpurun_out/r03_two_ranks_full.err | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/r03_two_ranks_full.json')); c=d['config']
print(d['n_gpus'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])
print(c['workload']); print(c['single_gpu_same_shard'], c['weak_scaling_vs_same_shard'])
w=c['c2_weak']; print('c2_weak', w['value'], w['ms_per_step'], w['n_variables_total'])
print(c['bytes_resident']); print('exchange', c['exchange'])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:160])
print('linear', c['linear_trial_mode']['fit_iterations_per_sec'], c['linear_trial_mode']['x_passes_per_iteration'])"
