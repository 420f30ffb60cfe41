"""CPU oracle for the Linear CorEx non-synergistic fit path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the algorithm on the hot path of the reference
(`/root/reference/linearcorex/linearcorex.py`, cited per function as `ref :LINE`).  It exists so that
the HIP implementation in `linearcorex_amd/` can be checked against something that runs on a
CPU-only box.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the product package never does (it fails loudly without its HIP library instead).

Pinning: the reference ships no tests or golden vectors for this path, so the oracle
is pinned against outputs of the reference itself, generated in the build container by
`tests/golden/make_golden*.py` (which import the reference read-only) and committed as `.npz`
fixtures under `tests/golden/` (g1-g11: big5, planted clusters, the config-2 shape, edge cases, 'outliers', missing values, the
stacking loop, the synergistic branch, predict / invert / 'empirical', transform(details=True), pick_n_hidden).  `tests/test_oracle_golden.py` checks every function below against
those fixtures for both working precisions:

  * dtype=float32  - "oracle-32", what the reference literally computes (it hard-casts, ref :108,:116)
  * dtype=float64  - "oracle-64", the same arithmetic lifted to double (summation-order stable;
                      the one the 1e-6 get_covariance() tolerance is defined against).

Layout convention here is the reference's: W and every "M x V" moment are (n_hidden, n_variables).
"""
from __future__ import annotations

import numpy as np

ANNEAL_BASE = 0.6          # ref :119
ANNEAL_STAGES = 6          # ref :119
WOLFE_C1 = 0.1             # ref :327
G_TAIL = 4.0               # ref :483


# --------------------------------------------------------------------------------------------------
# preprocessing (ref :397-429, :483-510)
# --------------------------------------------------------------------------------------------------
def squash_tails(z, t=G_TAIL):
    """ref :483-487 `g`: identity inside [-t, t], tanh-compressed beyond."""
    inner = np.clip(z, -t, t)
    return inner + np.tanh(z - inner)


def unsquash_tails(z, t=G_TAIL):
    """ref :490-494 `g_inv`."""
    inner = np.clip(z, -t, t)
    return inner + np.arctanh(np.clip(z - inner, -1 + 1e-10, 1 - 1e-10))


def impute_column_means(x, sentinel):
    """ref :497-510 `mean_impute`: sentinel/NaN cells -> mean of the observed cells of that column.

    Returns (filled copy, per-column observed counts)."""
    x = np.array(x, copy=True)
    if not np.isnan(sentinel):
        x = np.where(x == sentinel, np.nan, x)
    counts = np.empty(x.shape[1], dtype=np.int64)
    for c in range(x.shape[1]):
        col = x[:, c]
        ok = np.isfinite(col)
        col[np.isnan(col)] = np.mean(col[ok])
        counts[c] = int(ok.sum())
    return x, counts


def preprocess(x, theta=None, gaussianize="standard", missing_values=None):
    """ref :397-429.  theta=None means "fit" (estimate mean/std); returns (x_tilde, theta, n_obs)."""
    if missing_values is not None:
        x, n_obs = impute_column_means(x, missing_values)
    else:
        n_obs = len(x)
    if gaussianize == "standard":
        if theta is None:
            mu = np.mean(x, axis=0)
            sd = np.sqrt(np.sum((x - mu) ** 2, axis=0) / n_obs).clip(1e-10)     # ref :413
            theta = (mu, sd)
        x = (x - theta[0]) / theta[1]
    elif gaussianize == "outliers":
        if theta is None:
            theta = (np.mean(x, axis=0), np.std(x, axis=0, ddof=0).clip(1e-10))  # ref :420-421
        x = squash_tails((x - theta[0]) / theta[1])
    elif gaussianize == "empirical":
        from scipy.stats import norm, rankdata
        x = np.array([norm.ppf((rankdata(col) - 0.5) / len(col)) for col in x.T]).T      # ref :424-426
    elif gaussianize == "none":
        pass
    else:
        raise ValueError("oracle covers gaussianize in {'standard','outliers','empirical','none'}")
    return x, theta, n_obs


# --------------------------------------------------------------------------------------------------
# moment engine (ref :196-288)
# --------------------------------------------------------------------------------------------------
def latent_second_moment(x, w, eps, n_samples=None):
    """u_j = (1-eps^2) * sum_l (x w^T)_lj^2 / n_samples + eps^2 * sum_i w_ji^2   (ref :247-249, :226-228).
    n_samples: the reference divides by `self.n_samples`, the sample count of the FIT (ref :249) - for every x it is handed,
    so `transform(x_new, details=True)` (ref :392-394) evaluates a batch of another size with the fit's divisor.
    None = len(x), which is the same thing inside `fit`."""
    y = x.dot(w.T)
    ssq = np.einsum("lj,lj->j", y, y)
    uj = (1 - eps ** 2) * ssq / (x.shape[0] if n_samples is None else n_samples) + eps ** 2 * np.sum(w ** 2, axis=1)
    return y, uj


def norm(x, w, eps):
    """ref :215-228 `_norm` = sqrt(u_j)."""
    return np.sqrt(latent_second_moment(x, w, eps)[1])


def sig(x, u, eps):
    """ref :196-213 `_sig`: (Sigma_eps u^T)^T without forming Sigma; result is (m, nv)."""
    proj = x.T.dot(x.dot(u.T))
    return (1 - eps ** 2) * proj.T / x.shape[0] + eps ** 2 * u


def moments_ns(x, w, eps, quick=False, yscale=1.0, n_samples=None):
    """ref :236-288 `_calculate_moments_ns`.  Returns False when quick and max(u_j) >= 1 (ref :250).
    n_samples: `self.n_samples` of the reference (ref :249, :260), see latent_second_moment."""
    ns = x.shape[0] if n_samples is None else n_samples
    y, uj = latent_second_moment(x, w, eps, ns)
    mo = {"uj": uj}
    if quick and np.max(uj) >= 1.0:
        return False
    rho = (1 - eps ** 2) * x.T.dot(y).T / ns + eps ** 2 * w                 # ref :259-260
    ry = w.dot(rho.T)                                                       # ref :261
    mo["rho"] = rho
    mo["Y_j^2"] = yscale ** 2 / (1.0 - uj)                                  # ref :262
    np.fill_diagonal(ry, 1)                                                 # ref :263
    mo["ry"] = ry
    invrho = 1.0 / (1.0 - rho ** 2)
    rir = rho * invrho
    mo["invrho"], mo["rhoinvrho"] = invrho, rir
    qij = ry.dot(rir)                                                       # ref :266
    si = np.sum(rho * rir, axis=0)                                          # ref :268
    mo["Qij"], mo["Si"] = qij, si
    mo["Qi-Si^2"] = np.einsum("ki,ki->i", rir, qij - si * rho)              # ref :269
    mo["TC"] = (np.sum(np.log(1 + si))
                - 0.5 * np.sum(np.log(1 + mo["Qi-Si^2"]))
                + 0.5 * np.sum(np.log(1 - uj)))                             # ref :272-274
    if not quick:
        mi = -0.5 * np.log1p(-rho ** 2)                                     # ref :278
        mo["MI"] = mi
        mo["X_i Y_j"] = rho.T * np.sqrt(mo["Y_j^2"])                        # ref :279
        xz = np.linalg.solve(ry, rho).T                                     # ref :280
        mo["X_i Z_j"] = xz
        mo["X_i^2 | Y"] = (1.0 - np.einsum("ij,ji->i", xz, rho)).clip(1e-6)  # ref :281
        iyx = 0.5 * np.log(mo["Y_j^2"]) - 0.5 * np.log(yscale ** 2)         # ref :282
        ixy = -0.5 * np.log(mo["X_i^2 | Y"])                                # ref :283
        mo["I(Y_j ; X)"], mo["I(X_i ; Y)"] = iyx, ixy
        mo["TCs"] = mi.sum(axis=1) - iyx                                    # ref :284
        mo["TC_no_overlap"] = mi.max(axis=0).sum() - iyx.sum()              # ref :285
        mo["TC_direct"] = ixy.sum() - iyx                                   # ref :286
        mo["additivity"] = (mi.sum(axis=0) - ixy).sum()                     # ref :287
    return mo


# --------------------------------------------------------------------------------------------------
# one fixed-point iteration (ref :290-334)
# --------------------------------------------------------------------------------------------------
def update_direction(x, w, mo, eps):
    """Everything of ref :292-305 up to and including the tangent: returns a dict with
    H, grad, sig_grad, Bj, update, tangent."""
    rj = 1.0 - mo["uj"][:, np.newaxis]
    rir, rho, inv = mo["rhoinvrho"], mo["rho"], mo["invrho"]
    q2 = mo["Qi-Si^2"]
    h = np.dot(rir / (1 + q2), rir.T)                                       # ref :294
    np.fill_diagonal(h, 0)
    grad = w / rj                                                           # ref :296
    grad -= 2 * inv * rir / (1 + mo["Si"])                                  # ref :297
    grad += inv ** 2 * ((1 + rho ** 2) * mo["Qij"] - 2 * rho * mo["Si"]) / (1 + q2)   # ref :298-299
    grad += np.dot(h, w)                                                    # ref :300
    sg = sig(x, grad, eps)                                                  # ref :301
    bj = np.sum(rho * grad, axis=1, keepdims=True)                          # ref :302
    upd = -rj * (grad - 2.0 * w / (2 - rj) * bj)                            # ref :303
    tangent = np.einsum("ji,ji", sg, upd)                                   # ref :305
    return {"H": h, "grad": grad, "sig_grad": sg, "Bj": bj[:, 0], "update": upd, "tangent": tangent}


def update_ns(x, w, mo, eps, tol=1e-5):
    """ref :290-334 `_update_ns`.  Returns (w_new, moments_new, info).

    info: status in {"ok", "singular", "step_too_small"}, eta (last tried), n_trials (moment
    evaluations issued), n_invalid (trials that hit the u_j >= 1 exit), tangent."""
    d = update_direction(x, w, mo, eps)
    info = {"tangent": d["tangent"], "n_trials": 0, "n_invalid": 0, "eta": 0.0, "status": "ok"}
    if d["tangent"] >= 0:                                                   # ref :306-311
        info["status"] = "singular"
        return w, mo, info
    eta = 1.0
    w_try, m_try = w, mo
    while True:
        if eta < min(tol, 1e-10):                                           # ref :316-319
            info["status"] = "step_too_small"
            break
        w_try = w + eta * d["update"]                                       # ref :320
        m_try = moments_ns(x, w_try, eps, quick=True)
        info["n_trials"] += 1
        info["eta"] = eta
        if m_try is False:                                                  # ref :322-326
            info["n_invalid"] += 1
            eta *= 0.5
            continue
        if not (-m_try["TC"] <= -mo["TC"] + WOLFE_C1 * eta * d["tangent"]):  # ref :327-332
            eta *= 0.5
            continue
        break
    return w_try, m_try, info


def anneal_schedule(anneal=True, warm_start=False):
    """ref :113-119."""
    if anneal and not warm_start:
        return [ANNEAL_BASE ** k for k in range(1, ANNEAL_STAGES + 1)] + [0]
    return [0.0]


def rescale_for_stage(w, uj, eps_old, eps_new):
    """ref :129-133: keep u_j < 1 when the annealing parameter changes."""
    wmag = np.sum(w ** 2, axis=1, keepdims=True)
    delta = (eps_new ** 2 - eps_old ** 2) / (1.0 - eps_new ** 2) * wmag / uj.reshape((-1, 1))
    a = np.sqrt((1.0 - eps_old ** 2) / ((1.0 - eps_new ** 2) * (1.0 + delta)))
    return w * (0.001 * np.floor(1000.0 * a))


def initial_weights(seed, m, nv, dtype):
    """ref :89 + :116 - the legacy global RandomState stream, seeded in the constructor."""
    return np.random.RandomState(seed).randn(m, nv).astype(dtype)


def invert(x, theta, gaussianize="standard"):
    """ref :431-438 `invert`: undo the marginal preprocessing."""
    if gaussianize == "standard":
        return theta[1] * x + theta[0]
    if gaussianize == "outliers":
        return theta[1] * unsquash_tails(x) + theta[0]
    return x


def predict(xz, y, theta, gaussianize="standard"):
    """ref :440-441 `predict`: invert(X_i Z_j . y^T), xz = moments["X_i Z_j"] (nv, m), y (n, m) -> (n, nv)."""
    return invert(np.dot(xz, y.T).T, theta, gaussianize)


class OracleFit:
    """Result of `fit_ns` (mirrors the reference attributes the parity tests look at)."""

    def __init__(self):
        self.ws = None
        self.moments = None
        self.theta = None
        self.eps = 0
        self.history_tc = []
        self.n_moment_calls = 0
        self.n_trials = 0
        self.n_invalid = 0
        self.stage_iters = []
        self.x_tilde = None
        self.w_init = None

    # ref :193-194
    def clusters(self):
        return np.argmax(np.abs(self.ws), axis=0)

    # ref :443-455
    def get_covariance(self):
        if getattr(self, "synergistic", False):
            return covariance_syn(self.moments, self.theta)
        return covariance_ns(self.moments, self.eps, self.theta)

    # ref :386-395
    def transform(self, x_tilde):
        return x_tilde.dot(self.ws.T)


def covariance_ns(mo, eps, theta):
    """ref :443-451 (non-synergistic branch)."""
    z = mo["rhoinvrho"] / (1 + mo["Si"])
    cov = np.dot(z.T, z)
    cov /= (1.0 - eps ** 2)
    np.fill_diagonal(cov, 1)
    return theta[1][:, np.newaxis] * theta[1] * cov


def fit_ns(x, n_hidden, seed=0, max_iter=10000, tol=1e-5, anneal=True, dtype=np.float32,
           gaussianize="standard", missing_values=None, w0=None, keep_x=False, on_iteration=None):
    """ref :107-164 `fit` for discourage_overlap=True.

    w0 != None reproduces the warm-start path (skip init, schedule [0.], ref :113-119).
    on_iteration(stage, it, w, mo, info) is a test hook."""
    x = np.asarray(x, dtype=dtype)                                          # ref :108
    x, theta, _ = preprocess(x, None, gaussianize, missing_values)
    out = fit_ns_preprocessed(x, n_hidden, seed, max_iter, tol, anneal, dtype, w0, keep_x, on_iteration)
    out.theta = theta
    return out


def fit_ns_preprocessed(x, n_hidden, seed=0, max_iter=10000, tol=1e-5, anneal=True, dtype=np.float32,
                        w0=None, keep_x=False, on_iteration=None, finish=True):
    """The loop of ref :110-164 on an already preprocessed x (finish=False stops before the
    final detail moments / factor sort of ref :160-163 - the region bench.py times)."""
    out = OracleFit()
    ns, nv = x.shape
    if w0 is None:
        w = initial_weights(seed, n_hidden, nv, dtype)
        w /= (10.0 * norm(x, w, 0))[:, np.newaxis]                          # ref :117
        sched = anneal_schedule(anneal)
    else:
        w = np.array(w0, copy=True)
        sched = anneal_schedule(anneal, warm_start=True)
    out.w_init = w.copy()
    mo = moments_ns(x, w, 0, quick=True)                                    # ref :122
    eps = 0
    for stage, eps_new in enumerate(sched):
        eps_old, eps = eps, eps_new
        if stage > 0:
            w = rescale_for_stage(w, mo["uj"], eps_old, eps)
        mo = moments_ns(x, w, eps, quick=False)                             # ref :134
        n_it = 0
        for it in range(max_iter):
            last_tc = mo["TC"]
            w, mo, info = update_ns(x, w, mo, eps, tol)                     # ref :139
            out.n_trials += info["n_trials"]
            out.n_invalid += info["n_invalid"]
            n_it += 1
            if mo is False:                                                 # ref :144-149 (raises -> return)
                out.ws, out.moments, out.eps = w, mo, eps
                return out
            # a merely non-finite TC only prints in the reference and the loop continues (ref :146)
            delta = np.abs(mo["TC"] - last_tc)
            out.history_tc.append(mo["TC"])                                 # ref :151,:169
            if on_iteration is not None:
                on_iteration(stage, it, w, mo, info)
            if delta < tol:                                                 # ref :152
                break
        out.stage_iters.append(n_it)
    if not finish:
        out.ws, out.moments, out.eps = w, mo, eps
        return out
    mo = moments_ns(x, w, eps, quick=False)                                 # ref :160
    order = np.argsort(-mo["TCs"])                                          # ref :161
    w = w[order]
    mo = moments_ns(x, w, eps, quick=False)                                 # ref :163
    out.ws, out.moments, out.eps = w, mo, eps
    if keep_x:
        out.x_tilde = x
    return out


# --------------------------------------------------------------------------------------------------
# synthetic inputs used by tests and bench
# --------------------------------------------------------------------------------------------------
def gen_iid(n, v, seed=1, dtype=np.float64):
    """Gen-A: iid N(0,1)."""
    return np.random.RandomState(seed).randn(n, v).astype(dtype)


# --------------------------------------------------------------------------------------------------
# synergistic branch, discourage_overlap=False (ref :336-384, :119-121, :141, :452-455)
# --------------------------------------------------------------------------------------------------
SYN_ETA = 0.1              # ref :141


def moments_syn(x, w, yscale=1.0, n_samples=None):
    """ref :336-373 `_calculate_moments_syn` (the `quick` flag is ignored there).  n_samples: `self.n_samples` of the
    reference (ref :352, :355), the fit's sample count whatever x is (see latent_second_moment)."""
    n = x.shape[0] if n_samples is None else n_samples
    m = w.shape[0]
    mo = {}
    y = x.dot(w.T)                                                          # ref :347
    mo["X_i Y_j"] = x.T.dot(y) / n                                          # ref :355
    mo["cy"] = w.dot(mo["X_i Y_j"]) + yscale ** 2 * np.eye(m)               # ref :356
    mo["Y_j^2"] = np.diag(mo["cy"]).copy()                                  # ref :357
    sd = np.sqrt(mo["Y_j^2"])
    mo["ry"] = mo["cy"] / (sd * sd[:, np.newaxis])                          # ref :358
    mo["rho"] = (mo["X_i Y_j"] / sd).T                                      # ref :359
    mo["invrho"] = 1.0 / (1.0 - mo["rho"] ** 2)                             # ref :360
    mo["rhoinvrho"] = mo["rho"] * mo["invrho"]                              # ref :361
    mo["Qij"] = np.dot(mo["ry"], mo["rhoinvrho"])                           # ref :362
    mo["Qi"] = np.einsum("ki,ki->i", mo["rhoinvrho"], mo["Qij"])            # ref :363
    mo["Si"] = np.sum(mo["rho"] * mo["rhoinvrho"], axis=0)                  # ref :364
    mo["MI"] = -0.5 * np.log1p(-mo["rho"] ** 2)                             # ref :366
    mo["X_i Z_j"] = np.linalg.solve(mo["cy"], mo["X_i Y_j"].T).T            # ref :367
    mo["X_i^2 | Y"] = (1.0 - np.einsum("ij,ij->i", mo["X_i Z_j"], mo["X_i Y_j"])).clip(1e-6)   # ref :368
    iyx = 0.5 * np.log(mo["Y_j^2"]) - 0.5 * np.log(yscale ** 2)             # ref :369
    ixy = -0.5 * np.log(mo["X_i^2 | Y"])                                    # ref :370
    mo["TCs"] = mo["MI"].sum(axis=1) - iyx                                  # ref :371
    mo["additivity"] = (mo["MI"].sum(axis=0) - ixy).sum()                   # ref :372
    mo["TC"] = np.sum(ixy) - np.sum(iyx)                                    # ref :373
    return mo


def update_syn(x, w, mo, eta=SYN_ETA, yscale=1.0):
    """ref :375-384 `_update_syn`: damped fixed-point step, then fresh moments."""
    xz, x2y = mo["X_i Z_j"], mo["X_i^2 | Y"]
    h = (1.0 / x2y * xz.T).dot(xz)                                          # ref :378
    np.fill_diagonal(h, 0)
    r = xz.T / x2y                                                          # ref :380
    s = np.dot(h, w)                                                        # ref :381
    w_new = (1.0 - eta) * w + eta * (r - s)                                 # ref :382
    return w_new, moments_syn(x, w_new, yscale)


def covariance_syn(mo, theta):
    """ref :452-455 (synergistic branch of get_covariance)."""
    cov = np.einsum("ij,kj->ik", mo["X_i Z_j"], mo["X_i Y_j"])
    np.fill_diagonal(cov, 1)
    return theta[1][:, np.newaxis] * theta[1] * cov


def initial_weights_syn(seed, m, nv, yscale=1.0):
    """ref :89 + :121 - float64 draws, never cast: the reference's synergistic branch runs its W and
    moments in float64 whatever the dtype of x."""
    return np.random.RandomState(seed).randn(m, nv) * yscale ** 2 / np.sqrt(nv)


def fit_syn(x, n_hidden, seed=0, max_iter=10000, tol=1e-5, dtype=np.float32, gaussianize="standard",
            missing_values=None, keep_x=False):
    """ref :107-164 `fit` for discourage_overlap=False: no annealing (schedule [0.], :113-121), `_update_syn`
    with eta=0.1 until |dTC| < tol, then the TCs sort (:160-163).  dtype is the precision x is cast to and
    preprocessed in (ref :108); W stays float64."""
    x = np.asarray(x, dtype=dtype)
    xt, theta, _ = preprocess(x, None, gaussianize, missing_values)
    res = OracleFit()
    res.theta = theta
    w = initial_weights_syn(seed, n_hidden, xt.shape[1])
    res.w_init = w.copy()
    mo = moments_syn(xt, w)                                                 # ref :122
    mo = moments_syn(xt, w)                                                 # ref :134 (same weights)
    for _ in range(max_iter):                                               # ref :136-155
        last = mo["TC"]
        w, mo = update_syn(xt, w, mo)
        if not np.isfinite(mo["TC"]):
            break
        res.history_tc.append(mo["TC"])
        if np.abs(mo["TC"] - last) < tol:
            break
    mo = moments_syn(xt, w)                                                 # ref :160
    order = np.argsort(-mo["TCs"])                                          # ref :161
    w = w[order]                                                            # ref :162
    res.ws, res.moments = w, moments_syn(xt, w)                             # ref :163
    res.synergistic = True
    if keep_x:
        res.x_tilde = xt
    return res


def gen_planted(n, v, m, seed=1, noise=1.0, dtype=np.float64):
    """Gen-B: m planted groups; returns (X, group-of-each-variable)."""
    rng = np.random.RandomState(seed)
    z = rng.randn(n, m)
    grp = rng.randint(0, m, v)
    x = z[:, grp] + noise * rng.randn(n, v)
    return x.astype(dtype), grp
