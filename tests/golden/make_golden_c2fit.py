#!/usr/bin/env python3
"""Golden fixture of BASELINE.json configs[1] END TO END, produced by RUNNING THE REFERENCE here: synthetic Gaussian X
10 000 x 5 000 (RandomState(1).randn, the matrix tests/test_full_size_gpu.py and bench.py's c2 block draw), n_hidden = 32, the
float64-lifted reference (make_golden.precision), the whole fit with the reference's defaults (tol 1e-5 per annealing stage).
Stored: the TC history, the line-search trial counts, clusters, ws, TCs, and of the 5 000 x 5 000 covariance the diagonal, the
Frobenius norm and a block of 16 rows - arrays only.  About two minutes on 8 cores; the reference is imported, never copied.

Usage:  python tests/golden/make_golden_c2fit.py      (writes tests/golden/g12_c2_fit.npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import run_reference  # noqa: E402


def main():
    n, v, m = 10000, 5000, 32
    x = np.random.RandomState(1).randn(n, v)
    model, cov, yt, clusters = run_reference(x, "f64", m, seed=0)
    out = {"shape": np.array([n, v, m]),
           "history_tc": np.asarray(model.history["TC"], np.float64),
           "trials_per_iter": np.asarray(model.trials_per_iter, np.int32),
           "n_moment_calls": np.int64(model.n_moment_calls), "n_invalid": np.int64(model.n_invalid),
           "clusters": np.asarray(clusters, np.int64), "ws": np.asarray(model.ws, np.float64),
           "tcs": np.asarray(model.tcs, np.float64), "tc": np.float64(model.tc),
           "cov_diag": np.diag(cov).copy(), "cov_fro": np.float64(np.linalg.norm(cov)),
           "cov_rows": np.array([7, 8, 9, 10, 1000, 1001, 1002, 1003, 2499, 2500, 2501, 2502, 4996, 4997, 4998, 4999])}
    out["cov_block"] = cov[out["cov_rows"]].copy()
    out["transform_head"] = np.asarray(yt[:64], np.float64)
    np.savez_compressed(os.path.join(HERE, "g12_c2_fit.npz"), **out)
    print("g12_c2_fit.npz: %d iterations, %d trials, TC %.6f" % (len(out["history_tc"]), int(out["trials_per_iter"].sum()), float(out["tc"])))


if __name__ == "__main__":
    main()
