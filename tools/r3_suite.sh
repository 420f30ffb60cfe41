#!/bin/bash
# the whole GPU suite + smoke, timing per test file
mkdir -p gpurun_out/r3_suite
T0=$(date +%s)
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r3_suite/suite.log 2>&1
echo "suite rc=$? in $(( $(date +%s) - T0 )) s" | tee gpurun_out/r3_suite/summary.txt
tail -25 gpurun_out/r3_suite/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r3_suite/summary.txt
