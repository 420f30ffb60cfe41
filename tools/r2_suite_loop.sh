#!/bin/bash
# the GPU suite several times in a row: flaky tests (timing bars, multi-process launches) must show up here, not in the driver's run
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in $(seq 1 ${1:-6}); do
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tr '\n' ' '; echo
done | tee gpurun_out/r02_suite_loop.txt
