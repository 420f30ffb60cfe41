#!/usr/bin/env python3
"""Wall-clock of whole fits on small inputs (BASELINE config 1: test_big5, 2000 x 50, n_hidden=5), device vs the
NumPy oracle on the host - the latency-bound end of the path (GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from linearcorex_amd import Corex  # noqa: E402
from oracle import corex_oracle as O  # noqa: E402

g1 = np.load(os.path.join(ROOT, "tests", "golden", "g1_big5.npz"))
x = g1["x_raw"].astype(np.float64)
for dt in (np.float32, np.float64):
    for mode in ("exact", "linear"):
        Corex(n_hidden=5, seed=0, dtype=dt, line_search=mode).fit(x)          # warm up (library load, allocations)
        t0 = time.perf_counter()
        m = Corex(n_hidden=5, seed=0, dtype=dt, line_search=mode).fit(x)
        t1 = time.perf_counter()
        n = len(m.history["TC"])
        print("device %s %-6s: %4d iterations in %.3f s = %6.0f it/s  (%.0f us/iteration, %.2f trials/iteration) TC %.5f"
              % (np.dtype(dt).name, mode, n, t1 - t0, n / (t1 - t0), (t1 - t0) / n * 1e6, m.stats["trials"] / n, float(m.tc)))
    t0 = time.perf_counter()
    r = O.fit_ns(x, 5, seed=0, dtype=dt)
    t1 = time.perf_counter()
    print("oracle %s       : %4d iterations in %.3f s = %6.0f it/s" % (np.dtype(dt).name, len(r.history_tc), t1 - t0,
                                                                         len(r.history_tc) / (t1 - t0)))
