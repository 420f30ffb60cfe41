#!/bin/bash
# round-2 first GPU session: host facts for the CPU-baseline design, the two-rank stall under stress, the suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
{
  echo "nproc: $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null
  grep -E "MemTotal|MemAvailable" /proc/meminfo; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"
  python -c "import numpy as np; np.show_config()" 2>&1 | grep -i -E "openblas|mkl|blas" | head -5
  df -h /tmp /dev/shm | tail -2
} > gpurun_out/r02_host_facts.txt 2>&1
cat gpurun_out/r02_host_facts.txt
timeout 900 python tools/dist_stress.py --reps 25 --timeout 60 --dump-after 30 > gpurun_out/r02_dist_stress_parent.txt 2>&1
tail -5 gpurun_out/r02_dist_stress_parent.txt
timeout 600 python tools/dist_stress.py --reps 10 --timeout 60 --dump-after 30 --shape 300x6001x8 > gpurun_out/r02_dist_stress_wide.txt 2>&1
tail -3 gpurun_out/r02_dist_stress_wide.txt
for i in 1 2; do timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3; done
ls -la gpurun_out/dist_stall_stacks.log 2>/dev/null
