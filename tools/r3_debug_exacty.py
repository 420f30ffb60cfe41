import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import corex_oracle as O
from linearcorex_amd import Corex
from linearcorex_amd.preprocess import preprocess as pp
x, _ = O.gen_planted(700, 900, 6, seed=91)
xt = pp(x.astype(np.float64), None, "standard", None)[0]
out = {}
for mode in ("exact", "exact-y"):
    model = Corex(n_hidden=6, seed=0, dtype=np.float64, tol=0.0, device=0, line_search=mode)
    be = model._attach_shard(xt, 900)
    per = []
    for i_eps, eps in enumerate(model._init_weights()):
        model._begin_stage(i_eps, eps)
        for k in range(25):
            t0 = model.stats["trials"]; i0 = model.stats["invalid_trials"]
            model._iterate(more=k + 1 < 25)
            per.append((model.stats["trials"] - t0, model.stats["invalid_trials"] - i0, float(model.tc)))
    out[mode] = per
    be.close()
for k, (a, b) in enumerate(zip(out["exact"], out["exact-y"])):
    if a[:2] != b[:2] or abs(a[2] - b[2]) > 1e-9 * abs(a[2]):
        print(k, a, b)
print("totals", sum(a[0] for a in out["exact"]), sum(b[0] for b in out["exact-y"]))
