#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -k merged 2>&1 | tail -5
