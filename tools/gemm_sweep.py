#!/usr/bin/env python3
"""Sweep launch geometries of the two X-streaming GEMM kernels on a resident synthetic X (GPU box).

    python tools/gemm_sweep.py c2          # 10000 x 5000, m=32, f64
    python tools/gemm_sweep.py c3lite      # 50000 x 20000, m=64, f32
"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

SHAPES = {"c2": (10000, 5000, 32, np.float64), "c2m64": (10000, 5000, 64, np.float64),
          "c2m64f32": (10000, 5000, 64, np.float32), "c2f32": (10000, 5000, 32, np.float32),
          "c2m128": (10000, 5000, 128, np.float64), "c2m128f32": (10000, 5000, 128, np.float32),
          "c2m16": (10000, 5000, 16, np.float64), "c2m16f32": (10000, 5000, 16, np.float32),
          "mid64": (20000, 20000, 64, np.float64), "mid64f32": (20000, 20000, 64, np.float32),
          "mid32": (20000, 20000, 32, np.float64), "mid32f32": (20000, 20000, 32, np.float32),
          "tall": (100000, 500, 16, np.float64), "tallf32": (100000, 500, 16, np.float32),
          "small": (500, 2000, 8, np.float64), "c5": (448, 20000, 32, np.float64), "c5f32": (448, 20000, 32, np.float32), "c3lite": (50000, 20000, 64, np.float32),
          "c4lite": (50000, 20000, 128, np.float32), "c3": (50000, 100000, 64, np.float32)}


def main():
    ge.build()
    from tests.probe import ProbeBackend as HipBackend      # the lab build of the engine (tools/liblcx_probe.so)
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    if name in SHAPES:
        n, v, m, dt = SHAPES[name]
    else:                                   # "NxVxM:f32"
        dims, tag = name.split(":")
        n, v, m = (int(t) for t in dims.split("x"))
        dt = np.float32 if tag == "f32" else np.float64
    es = np.dtype(dt).itemsize
    gb = es * (n * v + m * v + n * m) / 1e9
    fl = 2.0 * n * v * m / 1e12
    combos = [(None, None)] + list(itertools.product([4, 8], [1, 2, 3, 4, 6]))
    if os.environ.get("SWEEP_DEFAULT_ONLY"):
        combos = [(None, None)]
    for kind, pre in ((0, "LCX_NT"), (1, "LCX_TN")):
        for kw, s in combos:
            for k in ("LCX_NT_KW", "LCX_NT_S", "LCX_TN_KW", "LCX_TN_S"):
                os.environ.pop(k, None)
            if kw is not None:
                os.environ[pre + "_KW"], os.environ[pre + "_S"] = str(kw), str(s)
            be = HipBackend(n, v, m, dt, 0)
            be.generate_x(1, 0, 1, 0)
            be.set_ws((np.random.RandomState(0).randn(m, v) * 0.01).astype(dt))
            be.moments_a(0)
            ms = be.bench_gemm(kind, 20)
            g = be.geometry()
            print("%s %s kw=%s S=%s (used kw=%d S=%d bpc=%d): %.1f us  %.0f GB/s  %.1f TF/s" % (
                name, "nt" if kind == 0 else "tn", kw, s,
                g["nt_waves" if kind == 0 else "tn_waves"], g["nt_split" if kind == 0 else "tn_split"],
                g["nt_blocks_per_cu" if kind == 0 else "tn_blocks_per_cu"], ms * 1e3, gb / ms * 1e3, fl / ms * 1e3),
                flush=True)
            be.close()


if __name__ == "__main__":
    main()
