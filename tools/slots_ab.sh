#!/bin/bash
# iteration-level A/B of the gemm_ct slot cap (kernel selection rule) over shard shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; OUT=gpurun_out/slots_ab.txt; : > $OUT
for WL in "$@"; do
  for CAP in ${CAPS:-6 32}; do
    if [ $CAP = default ]; then unset LCX_CT_MAX_SLOTS; else export LCX_CT_MAX_SLOTS=$CAP; fi; python bench.py --workload $WL --steps 35 --warmup 7 --cpu-iters-per-stage 0 --no-convergence --no-also-linear 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['config']['launch_geometry']; u=d['roofline']['use_sites']; print('$WL cap=$CAP it/s %.1f ms %.3f kernel %s nt_S %d tn_S %d nt_us %.1f tn_us %.1f' % (d['value'], d['ms_per_step'], d['roofline']['kernel'][:28], g['nt_split'], g['tn_split'], u['gemm_nt']['avg_us'], u['gemm_tn']['avg_us']))" >> $OUT
  done
done
cat $OUT
