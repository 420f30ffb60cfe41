#!/usr/bin/env python3
"""Would a hipGraph shorten the chain of small dependent launches?  One moment evaluation (8 launches at config 2)
issued directly vs captured once and replayed (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from tests.probe import ProbeBackend as HipBackend  # noqa: E402  (the lab build of the engine: tools/liblcx_probe.so)

for name, (n, v, m, dt) in {"c2": (10000, 5000, 32, np.float64), "c5": (400, 20000, 30, np.float64),
                            "big5": (2000, 50, 5, np.float64), "c2f32": (10000, 5000, 32, np.float32)}.items():
    be = HipBackend(n, v, m, dt, 0)
    be.generate_x(1, 0, 1, 0)
    w = (np.random.RandomState(0).randn(m, v) * 0.003).astype(dt)
    be.set_ws(w)
    be.moments_a(0); be.moments_b(0, 0.1, 0); be.moments_c(0)
    be.read_state(0)
    be.lib.lcx_set_ws                      # set 1 needs weights too: a trial at eta = 0 of a zero direction is not available here,
    import ctypes as C                     # so evaluate set 1 on a copy of the same weights through the trial buffer
    be.update_b(0.1); be.update_c(0.1); be.update_d(); be.make_trial(0.0)
    d, g = be.bench_graph(0.1, 50)
    print("%-6s one evaluation: direct %.1f us, hipGraph replay %.1f us" % (name, d * 1e3, g * 1e3), flush=True)
    be.close()
