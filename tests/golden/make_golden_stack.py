#!/usr/bin/env python3
"""Golden fixture for the hierarchical stacking recipe of the reference's command line
(vis_corex.py:530-545: layer k+1 is fitted on transform() of layer k; layer 0 gets missing_values).

Runs THE REFERENCE (imported from /root/reference, never copied) in the build container on
tests/data/test_big5.csv with layers 5,1 and adni_blood.csv with layers 6,2,1, in float32 (verbatim) and
float64 (in-memory lift, see make_golden.py), seed 0 per layer, and records inputs/outputs only.

Usage:  python tests/golden/make_golden_stack.py      (writes tests/golden/g7_stack.npz)
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import L, REF, load_csv, precision  # noqa: E402


def stack(x, layers, tag, missing, max_iter):
    """The loop of vis_corex.py:530-545 with a fixed seed."""
    res = []
    with precision(tag), contextlib.redirect_stdout(io.StringIO()):
        x_prev = x
        for l, layer in enumerate(layers):
            if l == 0:
                model = L.Corex(n_hidden=layer, missing_values=missing, discourage_overlap=True, max_iter=max_iter,
                                seed=0).fit(x)
            else:
                x_prev = res[-1].transform(x_prev)
                model = L.Corex(n_hidden=layer, discourage_overlap=True, max_iter=max_iter, seed=0).fit(x_prev)
            res.append(model)
    return res


def main():
    out = {}
    big5 = load_csv(os.path.join(REF, "tests/data/test_big5.csv"))
    adni = load_csv(os.path.join(REF, "tests/data/adni_blood.csv"), skip_first_col=True)
    for name, x, layers, max_iter in (("big5", big5, [5, 1], 10000), ("adni", adni, [6, 2, 1], 80)):
        out[name + "_layers"] = np.array(layers)
        out[name + "_max_iter"] = np.array(max_iter)
        for tag in ("f32", "f64"):
            models = stack(x.astype(np.float64), layers, tag, -1e6, max_iter)
            for l, mdl in enumerate(models):
                p = "%s_%s_l%d_" % (name, tag, l)
                out[p + "tc"] = np.float64(mdl.tc)
                out[p + "tcs"] = np.asarray(mdl.tcs, np.float64)
                out[p + "ws"] = np.asarray(mdl.ws)
                out[p + "n_iter"] = np.array(len(mdl.history["TC"]))
                out[p + "clusters"] = mdl.clusters().astype(np.int64)
            print(name, tag, [(float(m.tc), len(m.history["TC"])) for m in models])
    np.savez_compressed(os.path.join(HERE, "g7_stack.npz"), **out)


if __name__ == "__main__":
    main()
