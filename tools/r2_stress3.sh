#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python tools/dist_stress.py --reps 60 --timeout 90 --dump-after 60 > gpurun_out/r02_dist_stress_3.txt 2>&1
tail -4 gpurun_out/r02_dist_stress_3.txt
timeout 900 python tools/dist_stress.py --reps 30 --timeout 90 --dump-after 60 --mode linear --shape 300x6001x8 >> gpurun_out/r02_dist_stress_3.txt 2>&1
tail -2 gpurun_out/r02_dist_stress_3.txt
