#!/usr/bin/env python3
"""How far is a float32 device fit from the reference's own float32 run (golden fixtures)?  Prints the actual deviations
behind the float32 end-to-end bars of tests/test_parity_gpu.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linearcorex_amd import Corex
from oracle import corex_oracle as O

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, float(np.max(np.abs(b)))))

g1 = np.load(os.path.join(ROOT, "tests/golden/g1_big5.npz"))
out = Corex(n_hidden=5, seed=0, dtype=np.float32, device=0).fit(g1["x_raw"].astype(np.float64))
h, hr = np.asarray(out.history["TC"], np.float64), g1["f32_history_tc"]
print("big5 f32: iterations %d vs %d, final TC rel %.2e, cov rel %.2e, clusters equal %s, ws rel %.2e" % (
    len(h), len(hr), abs(h[-1] - hr[-1]) / abs(hr[-1]), rel(out.get_covariance(), g1["f32_cov"]),
    np.array_equal(out.clusters(), g1["f32_clusters"]), rel(out.ws, g1["f32_ws"]) if "f32_ws" in g1.files else -1))
n = min(len(h), len(hr)); print("   history rel over common prefix %.2e" % (np.max(np.abs(h[:n] - hr[:n]) / np.maximum(1, np.abs(hr[:n])))))
for name in ("g2_planted_small", "g2_planted_mid"):
    g = np.load(os.path.join(ROOT, "tests/golden/%s.npz" % name))
    nn, v, m = (int(t) for t in g["shape"])
    x, _ = O.gen_planted(nn, v, m)
    out = Corex(n_hidden=m, seed=0, dtype=np.float32, device=0).fit(x)
    h, hr = np.asarray(out.history["TC"], np.float64), g["f32_history_tc"]
    cov = out.get_covariance()
    print("%s f32: iterations %d vs %d, final TC rel %.2e, cov block rel %.2e, clusters equal %s" % (
        name, len(h), len(hr), abs(h[-1] - hr[-1]) / abs(hr[-1]), rel(cov[:256, :256], g["f32_cov_block"]),
        np.array_equal(out.clusters(), g["f32_clusters"])))
    n = min(len(h), len(hr)); print("   history rel over common prefix %.2e" % (np.max(np.abs(h[:n] - hr[:n]) / np.maximum(1, np.abs(hr[:n])))))
