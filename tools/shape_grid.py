#!/usr/bin/env python3
"""Both X passes at the production geometry over a grid of shard shapes (GPU box): looks for shapes the launch rules
handle badly.  Prints microseconds, GB/s and TF/s per pass and flags passes far from both rooflines."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ge.build()
    from tests.probe import ProbeBackend as HipBackend      # the lab build of the engine (tools/liblcx_probe.so)
    ns = [256, 1000, 4000, 16000, 60000]
    vs = [64, 500, 3000, 20000, 120000]
    ms = [(5, np.float32), (30, np.float32), (60, np.float32), (120, np.float32), (30, np.float64), (60, np.float64)]
    for m, dt in ms:
        es = np.dtype(dt).itemsize
        for n in ns:
            for v in vs:
                if n * v * es > 12e9 or n * v < 2e5:
                    continue
                be = HipBackend(n, v, m, dt, 0)
                be.generate_x(1, 0, 1, 0)
                be.set_ws((np.random.RandomState(0).randn(m, v) * 0.01).astype(dt))
                be.moments_a(0)
                g = be.geometry()
                gb = es * (n * v) / 1e9
                fl = 2.0 * n * v * g["m_pad"] / 1e12
                line = "%-3s m=%-3d n=%-6d v=%-6d" % ("f32" if es == 4 else "f64", m, n, v)
                for kind, nm in ((0, "nt"), (1, "tn")):
                    t = be.bench_gemm(kind, 10)
                    kn = be.kernel_name(kind).split("<")[0].replace("lcx::gemm_", "").replace("_kernel", "")
                    gbs, tfs = gb / t * 1e3, fl / t * 1e3
                    peak_t = 157.3 if es == 4 else 78.6
                    frac = max(gbs / 8000.0, tfs / peak_t)
                    flag = " <<<" if (frac < 0.25 and gb * 1e3 > 20) else ""
                    line += " | %s %-3s S=%-3d %8.1f us %5.0f GB/s %5.1f TF/s%s" % (
                        nm, kn, g["nt_split" if kind == 0 else "tn_split"], t * 1e3, gbs, tfs, flag)
                print(line, flush=True)
                be.close()


if __name__ == "__main__":
    main()
