#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

The reference (/root/reference, read-only) is imported, never copied: this script only records the
inputs it was given and the outputs it produced.  It cannot run on the GPU box (no /root/reference
there); the committed `.npz` files are what travels.

Two precisions are captured from the same reference source:
  f32 - the reference verbatim (it hard-casts to float32);
  f64 - the reference "lifted": the module's `np` name is swapped, in memory, for a proxy whose
        `float32` attribute is `numpy.float64`; nothing on disk is edited.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import contextlib
import csv
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):      # the reference prints a cudamat hint at import
    import linearcorex.linearcorex as L              # noqa: E402


class _NP64:
    """numpy proxy with float32 -> float64 (the in-memory lift)."""
    float32 = np.float64

    def __getattr__(self, k):
        return getattr(np, k)


@contextlib.contextmanager
def precision(tag):
    old = L.np
    if tag == "f64":
        L.np = _NP64()
    try:
        yield
    finally:
        L.np = old


QUICK_KEYS = ["uj", "rho", "ry", "Y_j^2", "invrho", "rhoinvrho", "Qij", "Si", "Qi-Si^2", "TC"]
DETAIL_KEYS = ["MI", "X_i Y_j", "X_i Z_j", "X_i^2 | Y", "I(Y_j ; X)", "I(X_i ; Y)", "TCs",
               "TC_no_overlap", "TC_direct", "additivity"]


def key_name(k):
    return k.replace(" ", "_").replace("^", "p").replace("|", "g").replace(";", "s") \
            .replace("(", "").replace(")", "").replace("-", "m")


class Recorder(L.Corex):
    """The reference class with taps on its private hot-path methods (inputs/outputs only)."""

    def __init__(self, *a, capture_iters=(), **kw):
        super().__init__(*a, **kw)
        self.capture_iters = set(capture_iters)
        self.cap = {}
        self._it = 0
        self._in_update = False
        self._cur = None
        self.n_moment_calls = 0
        self.n_invalid = 0
        self.trials_per_iter = []
        self.stage_first_iter = []
        self.x_tilde = None
        self.w_init = None

    def _calculate_moments_ns(self, x, ws, quick=False):
        if self.x_tilde is None:
            self.x_tilde = np.array(x, copy=True)
            self.w_init = np.array(ws, copy=True)
        out = super()._calculate_moments_ns(x, ws, quick=quick)
        self.n_moment_calls += 1
        if self._in_update:
            self._trials += 1
            if out is False:
                self.n_invalid += 1
        return out

    def _sig(self, x, u):
        out = super()._sig(x, u)
        if self._cur is not None:
            self._cur["grad"] = np.array(u, copy=True)
            self._cur["sig_grad"] = np.array(out, copy=True)
        return out

    def _update_ns(self, x):
        rec = self._it in self.capture_iters
        if rec:
            self._cur = {"eps": float(self.eps), "w_in": np.array(self.ws, copy=True)}
            for k in QUICK_KEYS:
                self._cur["in_" + key_name(k)] = np.array(self.moments[k], copy=True)
        self._in_update, self._trials = True, 0
        w, m = super()._update_ns(x)
        self._in_update = False
        self.trials_per_iter.append(self._trials)
        if rec:
            self._cur["w_out"] = np.array(w, copy=True)
            self._cur["n_trials"] = self._trials
            if m:
                for k in QUICK_KEYS:
                    self._cur["out_" + key_name(k)] = np.array(m[k], copy=True)
            self.cap[self._it] = self._cur
            self._cur = None
        self._it += 1
        return w, m


def run_reference(x, tag, n_hidden, seed=0, capture_iters=(), **kw):
    with precision(tag), contextlib.redirect_stdout(io.StringIO()) as so:
        model = Recorder(n_hidden=n_hidden, seed=seed, capture_iters=capture_iters, **kw)
        model.fit(x)
        cov = model.get_covariance() if x.shape[1] <= 6000 else None
        yt = model.transform(x)
        clusters = model.clusters()
    model.stdout = so.getvalue()
    return model, cov, yt, clusters


def store(out, name, arr, thin=None, nv=None):
    """Store arr; if it has a variable axis (length nv) and thin is set, keep every thin-th variable
    plus the Frobenius norm of the full array."""
    arr = np.asarray(arr)
    if thin and nv and arr.ndim >= 1 and nv in arr.shape and arr.size > 4096:
        ax = list(arr.shape).index(nv)
        sl = [slice(None)] * arr.ndim
        sl[ax] = slice(None, None, thin)
        out[name + "_thin"] = arr[tuple(sl)].copy()
        out[name + "_fro"] = np.float64(np.linalg.norm(arr.astype(np.float64)))
    else:
        out[name] = arr


def pack_fit(prefix, model, cov, yt, clusters, out, cov_block=None, thin=None):
    p = prefix
    nv = model.ws.shape[1]
    out[p + "history_tc"] = np.asarray(model.history["TC"], dtype=np.float64)
    out[p + "trials_per_iter"] = np.asarray(model.trials_per_iter, dtype=np.int32)
    out[p + "n_moment_calls"] = np.int64(model.n_moment_calls)
    out[p + "n_invalid"] = np.int64(model.n_invalid)
    store(out, p + "ws", model.ws, thin, nv)
    out[p + "clusters"] = clusters.astype(np.int64)
    out[p + "tc"] = np.float64(model.tc)
    out[p + "tcs"] = np.asarray(model.tcs)
    out[p + "theta_mean"], out[p + "theta_std"] = model.theta
    store(out, p + "w_init", model.w_init, thin, nv)
    for k in QUICK_KEYS + DETAIL_KEYS:
        store(out, p + "mom_" + key_name(k), model.moments[k], thin, nv)
    if cov is not None:
        out[p + "cov_fro"] = np.float64(np.linalg.norm(cov.astype(np.float64)))
        out[p + "cov_diag"] = np.diag(cov).copy()
        if cov_block is None:
            out[p + "cov"] = cov
        else:
            out[p + "cov_block"] = cov[:cov_block, :cov_block].copy()
            out[p + "cov_lastrows"] = cov[-4:, :].copy()
    if thin:
        out[p + "transform_thin"] = yt[::thin].copy()
        out[p + "transform_fro"] = np.float64(np.linalg.norm(yt.astype(np.float64)))
    else:
        out[p + "transform"] = yt
    for it, c in model.cap.items():
        for k, v in c.items():
            store(out, p + "step%d_%s" % (it, k), v, thin, nv)


def load_csv(path, skip_header=True, skip_first_col=False, delimiter=","):
    """Same recipe as the reference CLI (vis_corex.py:496-512): csv.reader in text mode."""
    with open(path, "r") as f:
        rows = list(csv.reader(f, delimiter=delimiter))
    if skip_header:
        rows = rows[1:]
    if skip_first_col:
        rows = [r[1:] for r in rows]
    return np.array(rows, dtype=float)


def planted(n, v, m, seed=1, noise=1.0):
    """Gen-B of SURVEY.md §8d (kept in sync with oracle.corex_oracle.gen_planted)."""
    rng = np.random.RandomState(seed)
    z = rng.randn(n, m)
    grp = rng.randint(0, m, v)
    return z[:, grp] + noise * rng.randn(n, v), grp


def main():
    # ---- G1: big5 (config 1) -------------------------------------------------------------------
    big5 = load_csv(os.path.join(REF, "tests/data/test_big5.csv"))
    out = {"x_raw": big5.astype(np.float32)}           # values are small integers: exact in f32
    for tag in ("f32", "f64"):
        model, cov, yt, cl = run_reference(big5, tag, 5, seed=0, capture_iters=(0, 1, 160, 230))
        pack_fit(tag + "_", model, cov, yt, cl, out)
        out[tag + "_x_tilde"] = model.x_tilde
        print("G1 big5", tag, "iters", len(model.history["TC"]), "TC", float(model.tc),
              "moment calls", model.n_moment_calls)
    np.savez_compressed(os.path.join(HERE, "g1_big5.npz"), **out)

    # ---- G2: planted clusters, two sizes ---------------------------------------------------------
    for name, (n, v, m, thin) in {"g2_planted_small": (500, 2000, 8, 8), "g2_planted_mid": (2000, 5000, 16, 20)}.items():
        x, grp = planted(n, v, m)
        out = {"shape": np.array([n, v, m]), "grp": grp.astype(np.int64)}
        for tag in ("f32", "f64"):
            model, cov, yt, cl = run_reference(x, tag, m, seed=0, capture_iters=(0, 5))
            pack_fit(tag + "_", model, cov, yt, cl, out, cov_block=256, thin=thin)
            print(name, tag, "iters", len(model.history["TC"]), "TC", float(model.tc),
                  "mean trials", np.mean(model.trials_per_iter), "invalid", model.n_invalid)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)

    # ---- G3: step level on a config-2-shaped input (N=10^4, V=5*10^3, M=32) ----------------------
    n, v, m = 10000, 5000, 32
    x = np.random.RandomState(1).randn(n, v)
    out = {"shape": np.array([n, v, m])}
    for tag in ("f32", "f64"):
        with precision(tag), contextlib.redirect_stdout(io.StringIO()):
            model = Recorder(n_hidden=m, seed=0, max_iter=2, capture_iters=(0, 1, 2))
            model.fit(x)
        p = tag + "_"
        out[p + "history_tc"] = np.asarray(model.history["TC"], dtype=np.float64)
        out[p + "trials_per_iter"] = np.asarray(model.trials_per_iter, dtype=np.int32)
        for it, c in model.cap.items():
            for k, val in c.items():
                store(out, p + "step%d_%s" % (it, k), val, 25, v)
        print("G3", tag, "TC history", out[p + "history_tc"][:4], "trials", model.trials_per_iter[:6])
    np.savez_compressed(os.path.join(HERE, "g3_c2_step.npz"), **out)

    # ---- G4: edge cases ---------------------------------------------------------------------------
    out = {}
    rng = np.random.RandomState(7)
    x = rng.randn(300, 40)
    x = (x - x.mean(0)) / x.std(0)
    w = rng.randn(4, 40)
    for tag in ("f32", "f64"):
        with precision(tag), contextlib.redirect_stdout(io.StringIO()):
            model = L.Corex(n_hidden=4, seed=0)
            model.n_samples, model.nv = x.shape
            dt = np.float32 if tag == "f32" else np.float64
            xs, wbig = x.astype(dt), (w * 3.0).astype(dt)
            res = model._calculate_moments_ns(xs, wbig, quick=True)            # max uj >= 1 -> False
            out[tag + "_invalid_is_false"] = np.bool_(res is False)
            full = model._calculate_moments_ns(xs, wbig, quick=False)          # non-quick never exits
            out[tag + "_invalid_uj"] = full["uj"]
            wsmall = (w * 0.02).astype(dt)
            model.eps = 0.36
            ok = model._calculate_moments_ns(xs, wsmall, quick=False)
            for k in QUICK_KEYS + DETAIL_KEYS:
                out[tag + "_eps036_" + key_name(k)] = np.asarray(ok[k])
            out[tag + "_eps036_sig"] = model._sig(xs, wsmall)
            out[tag + "_eps036_norm"] = model._norm(xs, wsmall)
    out["x"], out["w"] = x, w
    # duplicated columns: the reference's "nearly singular" warning path (tangent >= 0), if it triggers
    xd = rng.randn(200, 6)
    xd = np.concatenate([xd, xd[:, :3], xd[:, :3]], axis=1)
    out["dup_x"] = xd
    for tag in ("f32", "f64"):
        model, cov, yt, cl = run_reference(xd, tag, 3, seed=0, max_iter=300)
        out[tag + "_dup_history_tc"] = np.asarray(model.history["TC"], dtype=np.float64)
        out[tag + "_dup_ws"] = model.ws
        out[tag + "_dup_warned"] = np.bool_("nearly singular" in model.stdout)
        print("G4 dup", tag, "iters", len(model.history["TC"]), "warned", bool(out[tag + "_dup_warned"]))
    np.savez_compressed(os.path.join(HERE, "g4_edges.npz"), **out)

    # ---- G5: gaussianize='outliers' on a heavy-tailed stand-in for config 5 ----------------------
    n, v, m = 300, 1200, 6
    x, grp = planted(n, v, m, seed=3)
    heavy = np.arange(v) % 20 == 0
    x[:, heavy] = np.sign(x[:, heavy]) * np.abs(x[:, heavy]) ** 1.5
    out = {"shape": np.array([n, v, m]), "grp": grp.astype(np.int64)}
    for tag in ("f32", "f64"):
        model, cov, yt, cl = run_reference(x, tag, m, seed=0, gaussianize="outliers", capture_iters=(0,))
        pack_fit(tag + "_", model, cov, yt, cl, out, cov_block=128, thin=10)
        out[tag + "_x_tilde_thin"] = model.x_tilde[::10, ::10].copy()
        print("G5 outliers", tag, "iters", len(model.history["TC"]), "TC", float(model.tc))
    np.savez_compressed(os.path.join(HERE, "g5_outliers.npz"), **out)

    # ---- G6: missing values (adni_blood.csv, -1e6 sentinel; README recipe) -----------------------
    adni = load_csv(os.path.join(REF, "tests/data/adni_blood.csv"), skip_first_col=True)
    out = {"x_raw": adni.astype(np.float64)}
    for tag in ("f32", "f64"):
        model, cov, yt, cl = run_reference(adni, tag, 6, seed=0, missing_values=-1e6, max_iter=60)
        pack_fit(tag + "_", model, cov, yt, cl, out)
        out[tag + "_n_obs"] = np.asarray(model.n_obs)
        print("G6 adni", tag, "iters", len(model.history["TC"]), "TC", float(model.tc))
    np.savez_compressed(os.path.join(HERE, "g6_adni_missing.npz"), **out)


if __name__ == "__main__":
    main()
