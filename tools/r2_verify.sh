#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -c "import __graft_entry__ as g; g.build(); print('src hash', g._src_hash()[:16])"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
