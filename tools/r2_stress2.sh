#!/bin/bash
# round-2 stall hunt, part 2: the two-rank launches in their suite context (the files pytest runs before them), repeated,
# then under CPU pressure (the box gives 16 CPUs of quota on 256 logical CPUs: 48 busy loops keep the cgroup throttled)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/r02_dist_stress_2.txt
echo "# suite-context repeats: pytest tests/test_bench_gpu.py -k 'two_ranks or without_a_launcher' tests/test_cli.py tests/test_distributed_gpu.py" > $OUT
for i in $(seq 1 10); do
  T0=$(date +%s)
  timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_cli.py tests/test_distributed_gpu.py -x -q -m gpu -k "not default_line" 2>&1 | tail -1 >> $OUT
  echo "  repeat $i: $(( $(date +%s) - T0 )) s" >> $OUT
done
echo "# under CPU pressure (48 busy loops)" >> $OUT
for k in $(seq 1 48); do ( while :; do :; done ) & done
sleep 1
timeout 900 python tools/dist_stress.py --reps 15 --timeout 120 --dump-after 90 >> $OUT 2>&1
kill $(jobs -p) 2>/dev/null
wait 2>/dev/null
cat $OUT | tail -45
ls -la gpurun_out/dist_stall_stacks.log 2>/dev/null || echo "no stall log"
