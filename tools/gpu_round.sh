#!/bin/bash
# One GPU-box session: parity tests, smoke, the bench line, rocprofv3 kernel stats and PMC traffic of
# the same command.  Run through gpurun from the repo root; everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
WL=${2:-c2}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --workload $WL 2>gpurun_out/bench_$WL.err | tail -1 > gpurun_out/${TAG}_bench_$WL.json
cut -c1-900 gpurun_out/${TAG}_bench_$WL.json
bash $R/tools/gpu_prof.sh $TAG $WL
