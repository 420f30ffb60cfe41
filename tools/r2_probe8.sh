#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_edge_shapes_gpu.py -q -m gpu 2>&1 | tail -40
