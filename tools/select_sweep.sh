#!/bin/bash
# gemm_tn (/tn4) vs gemm_ct at the default geometry over a set of shard shapes: input to the kernel selection rule
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/select_sweep.txt; : > $OUT
for SH in "$@"; do
  for G in tn ct; do
    LCX_GEMM=$G SWEEP_DEFAULT_ONLY=1 python tools/gemm_sweep.py $SH 2>&1 | sed "s/^/$G /" >> $OUT
  done
done
cat $OUT
