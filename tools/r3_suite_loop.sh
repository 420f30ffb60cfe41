#!/bin/bash
# the GPU suite several times in a row on one box (flakiness check of the multi-rank / timing tests)
mkdir -p gpurun_out/r3_suite_loop
for k in 1 2 3; do
  T0=$(date +%s)
  python -m pytest tests -q -m gpu -x > gpurun_out/r3_suite_loop/suite_$k.log 2>&1
  echo "suite $k rc=$? in $(( $(date +%s) - T0 )) s: $(tail -1 gpurun_out/r3_suite_loop/suite_$k.log)" | tee -a gpurun_out/r3_suite_loop/summary.txt
done
