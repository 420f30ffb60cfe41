#!/bin/bash
# two ranks on GPU 0 through gloo (what tests/test_distributed_gpu.py launches), repeated, with the workers' output kept
#   bash tools/dist_debug.sh <mode> <timeout s> <repeats>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/dd
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for it in $(seq 1 ${3:-5}); do
  PORT=$((29617 + it))
  T0=$(date +%s)
  for r in 0 1; do
    RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=2 OPENBLAS_NUM_THREADS=2 LCX_TEST_DUMP_AFTER=40 LCX_CHECK_RANKS=1 \
      timeout ${2:-90} python tests/_dist_worker.py gpurun_out/dd 400 331 5 ${1:-exact} hip 25 > gpurun_out/dd/rank$r.$it.log 2>&1 &
  done
  wait
  T1=$(date +%s)
  echo "run $it: $((T1 - T0)) s; rank logs: $(wc -l < gpurun_out/dd/rank0.$it.log) $(wc -l < gpurun_out/dd/rank1.$it.log) lines"
  if [ $(wc -l < gpurun_out/dd/rank0.$it.log) -gt 5 ]; then tail -60 gpurun_out/dd/rank0.$it.log; tail -60 gpurun_out/dd/rank1.$it.log; fi
done
