#!/usr/bin/env python3
"""PCIe-inclusive rate of making a HOST matrix resident: lcx_upload_preprocess('standard') of an n x v float32 matrix (upload +
the three preprocess passes + the layout step), panel-major copy vs row-major + transposed copy.  python tools/upload_rate.py [n v]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n, v):
    import numpy as np
    from linearcorex_amd.backend import HipBackend
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, v), dtype=np.float32)
    be = HipBackend(n, v, 64, np.float32, 0)
    be.upload_preprocess(x[:, :], "standard")          # warm (allocations, first touch)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        be.upload_preprocess(x, "standard")
        ts.append(time.perf_counter() - t0)
    lay = be.bytes_resident()["x_layout"]
    t = min(ts)
    print("%-28s %d x %d float32 (%.1f GB): %.3f s = %.1f GB/s host -> resident, preprocessed (%s)"
          % (os.environ.get("LCX_X_LAYOUT", "auto"), n, v, x.nbytes / 1e9, t, x.nbytes / t / 1e9, lay), flush=True)
    be.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        n, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50000, 50000)
        for lay in ("panel", "rows"):
            subprocess.call([sys.executable, os.path.abspath(__file__), "child", str(n), str(v)], env=dict(os.environ, LCX_X_LAYOUT=lay, LCX_GEMM="ct"))
