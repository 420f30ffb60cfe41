"""Pin the CPU oracle (oracle/corex_oracle.py) to outputs of the reference itself.

The fixtures were produced by tests/golden/make_golden.py, which imports /root/reference in the
build container.  The oracle restates the same NumPy arithmetic in the same order, so on the same
NumPy/BLAS build it is expected to match bit for bit; tolerances below leave room for a different
BLAS on another host (summation order), nothing more.
"""
import numpy as np
import pytest

from oracle import corex_oracle as O
from tests.conftest import load_golden

DT = {"f32": np.float32, "f64": np.float64}
# different-BLAS headroom; on the generating host every comparison below is exact
RTOL = {"f32": 2e-4, "f64": 1e-9}


def key_name(k):
    return k.replace(" ", "_").replace("^", "p").replace("|", "g").replace(";", "s") \
            .replace("(", "").replace(")", "").replace("-", "m")


QUICK = ["uj", "rho", "ry", "Y_j^2", "invrho", "rhoinvrho", "Qij", "Si", "Qi-Si^2", "TC"]
DETAIL = ["MI", "X_i Y_j", "X_i Z_j", "X_i^2 | Y", "I(Y_j ; X)", "I(X_i ; Y)", "TCs",
          "TC_no_overlap", "TC_direct", "additivity"]


def close(a, b, tag, scale=1.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = RTOL[tag] * scale
    assert a.shape == b.shape
    assert np.allclose(a, b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max()) if b.size else 1.0)), \
        float(np.abs(a - b).max())


def thin_of(arr, g, name, nv):
    """Compare helper for thinned fixtures: returns (ours thinned the same way, golden)."""
    arr = np.asarray(arr)
    if name in g.files:
        return arr, g[name]
    gold = g[name + "_thin"]
    ax = list(arr.shape).index(nv)
    step = int(round(arr.shape[ax] / gold.shape[ax]))
    sl = [slice(None)] * arr.ndim
    sl[ax] = slice(None, None, step)
    return arr[tuple(sl)], gold


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_big5_end_to_end(g1, tag):
    r = O.fit_ns(g1["x_raw"].astype(np.float64), 5, seed=0, dtype=DT[tag], keep_x=True)
    assert len(r.history_tc) == len(g1[tag + "_history_tc"]) == {"f32": 263, "f64": 261}[tag]
    close(r.history_tc, g1[tag + "_history_tc"], tag)
    close(r.x_tilde, g1[tag + "_x_tilde"], tag)
    close(r.w_init, g1[tag + "_w_init"], tag)
    close(r.ws, g1[tag + "_ws"], tag, 50)
    assert np.array_equal(r.clusters(), g1[tag + "_clusters"])
    assert np.array_equal(r.clusters(), np.array([0, 2, 4, 1, 3])[np.arange(50) % 5])
    close(r.get_covariance(), g1[tag + "_cov"], tag, 50)
    close(r.transform(r.x_tilde), g1[tag + "_transform"], tag, 50)
    for k in QUICK + DETAIL:
        close(r.moments[k], g1[tag + "_mom_" + key_name(k)], tag, 50)
    assert abs(float(r.moments["TC"]) - 8.2089) < 1e-3
    assert r.n_trials + 10 == int(g1[tag + "_n_moment_calls"])     # 1 init + 7 stage starts + 2 final


def test_moment_call_accounting(g1):
    # reference issues: 1 (init, ref :122) + 7 (stage starts, :134) + 2 (final, :160/:163) + trials
    # + 0 from _norm (it does its own GEMM).  n_moment_calls counts _calculate_moments_ns only.
    for tag in ("f32", "f64"):
        trials = int(g1[tag + "_trials_per_iter"].sum())
        assert trials + 10 == int(g1[tag + "_n_moment_calls"])


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("it", [0, 1, 160, 230])
def test_big5_step_level(g1, tag, it):
    p = "%s_step%d_" % (tag, it)
    x = g1[tag + "_x_tilde"]
    w = g1[p + "w_in"]
    eps = float(g1[p + "eps"])
    mo = {k: g1[p + "in_" + key_name(k)] for k in QUICK}
    # moments at w_in reproduce the captured ones
    mine = O.moments_ns(x, w, eps, quick=True)
    for k in QUICK:
        close(mine[k], mo[k], tag, 10)
    d = O.update_direction(x, w, mo, eps)
    close(d["grad"], g1[p + "grad"], tag, 10)
    close(d["sig_grad"], g1[p + "sig_grad"], tag, 10)
    w2, m2, info = O.update_ns(x, w, mo, eps)
    assert info["n_trials"] == int(g1[p + "n_trials"])
    close(w2, g1[p + "w_out"], tag, 10)
    close(m2["TC"], g1[p + "out_TC"], tag, 10)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("which", ["g2_small", "g2_mid"])
def test_planted(request, tag, which):
    g = request.getfixturevalue(which)
    n, v, m = (int(t) for t in g["shape"])
    if which == "g2_mid" and tag == "f32":
        pytest.skip("covered by f64 at this size; keeps the CPU suite short")
    x, grp = O.gen_planted(n, v, m)
    assert np.array_equal(grp, g["grp"])
    r = O.fit_ns(x, m, seed=0, dtype=DT[tag], keep_x=True)
    assert len(r.history_tc) == len(g[tag + "_history_tc"])
    close(r.history_tc, g[tag + "_history_tc"], tag, 10)
    assert np.array_equal(r.clusters(), g[tag + "_clusters"])
    # planted structure is recovered exactly (purity 1): same cluster <=> same group
    cl = r.clusters()
    assert all(len(set(cl[grp == k])) == 1 for k in range(m))
    a, b = thin_of(r.ws, g, tag + "_ws", v)
    close(a, b, tag, 100)
    cov = r.get_covariance()
    close(cov[:256, :256], g[tag + "_cov_block"], tag, 100)
    close(cov[-4:], g[tag + "_cov_lastrows"], tag, 100)
    close(np.linalg.norm(cov.astype(np.float64)), g[tag + "_cov_fro"], tag, 100)
    assert r.n_trials == int(g[tag + "_trials_per_iter"].sum())
    assert r.n_invalid == int(g[tag + "_n_invalid"])
    for k in QUICK + DETAIL:
        a, b = thin_of(r.moments[k], g, tag + "_mom_" + key_name(k), v)
        close(a, b, tag, 100)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_c2_shaped_steps(g3, tag):
    n, v, m = (int(t) for t in g3["shape"])
    x = O.gen_iid(n, v, seed=1, dtype=np.float64)
    seen = []

    def hook(stage, it, w, mo, info):
        seen.append((stage, it, float(mo["TC"]), info["n_trials"]))

    r = O.fit_ns(x, m, seed=0, dtype=DT[tag], max_iter=2, on_iteration=hook)
    close(r.history_tc, g3[tag + "_history_tc"], tag, 10)
    assert [s[3] for s in seen] == list(g3[tag + "_trials_per_iter"])
    # step 0: captured inputs -> direction
    p = tag + "_step0_"
    xt = O.preprocess(x.astype(DT[tag]))[0]
    w = O.initial_weights(0, m, v, DT[tag])
    w /= (10.0 * O.norm(xt, w, 0))[:, np.newaxis]
    a, b = thin_of(w, g3, p + "w_in", v)
    close(a, b, tag, 10)
    eps = float(g3[p + "eps"])
    mo = O.moments_ns(xt, w, eps, quick=True)
    close(mo["uj"], g3[p + "in_uj"], tag, 10)
    close(mo["TC"], g3[p + "in_TC"], tag, 10)
    d = O.update_direction(xt, w, mo, eps)
    a, b = thin_of(d["grad"], g3, p + "grad", v)
    close(a, b, tag, 10)
    close(np.linalg.norm(d["sig_grad"].astype(np.float64)), g3[p + "sig_grad_fro"], tag, 10)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_edges(g4, tag):
    dt = DT[tag]
    x, w = g4["x"].astype(dt), g4["w"]
    assert bool(g4[tag + "_invalid_is_false"])
    assert O.moments_ns(x, (w * 3.0).astype(dt), 0, quick=True) is False       # ref :250-251
    with np.errstate(all="ignore"):
        full = O.moments_ns(x, (w * 3.0).astype(dt), 0, quick=False)
    close(full["uj"], g4[tag + "_invalid_uj"], tag)
    assert full["uj"].max() >= 1
    ws = (w * 0.02).astype(dt)
    mo = O.moments_ns(x, ws, 0.36, quick=False)
    for k in QUICK + DETAIL:
        close(mo[k], g4[tag + "_eps036_" + key_name(k)], tag, 10)
    close(O.sig(x, ws, 0.36), g4[tag + "_eps036_sig"], tag, 10)
    close(O.norm(x, ws, 0.36), g4[tag + "_eps036_norm"], tag, 10)
    # duplicated columns (near-singular covariance)
    r = O.fit_ns(g4["dup_x"], 3, seed=0, dtype=dt, max_iter=300)
    assert len(r.history_tc) == len(g4[tag + "_dup_history_tc"])
    close(r.history_tc, g4[tag + "_dup_history_tc"], tag, 100)
    close(r.ws, g4[tag + "_dup_ws"], tag, 1000)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_outliers(g5, tag):
    n, v, m = (int(t) for t in g5["shape"])
    x, grp = O.gen_planted(n, v, m, seed=3)
    heavy = np.arange(v) % 20 == 0
    x[:, heavy] = np.sign(x[:, heavy]) * np.abs(x[:, heavy]) ** 1.5
    r = O.fit_ns(x, m, seed=0, dtype=DT[tag], gaussianize="outliers", keep_x=True)
    close(r.x_tilde[::10, ::10], g5[tag + "_x_tilde_thin"], tag)
    assert len(r.history_tc) == len(g5[tag + "_history_tc"])
    close(r.history_tc, g5[tag + "_history_tc"], tag, 10)
    assert np.array_equal(r.clusters(), g5[tag + "_clusters"])
    close(r.get_covariance()[:128, :128], g5[tag + "_cov_block"], tag, 100)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_missing_values(g6, tag):
    x = g6["x_raw"]
    r = O.fit_ns(x, 6, seed=0, dtype=DT[tag], missing_values=-1e6, max_iter=60, keep_x=True)
    xt, theta, n_obs = O.preprocess(np.asarray(x, DT[tag]), None, "standard", -1e6)
    assert np.array_equal(n_obs, g6[tag + "_n_obs"])
    close(theta[0], g6[tag + "_theta_mean"], tag)
    close(theta[1], g6[tag + "_theta_std"], tag)
    assert len(r.history_tc) == len(g6[tag + "_history_tc"])
    close(r.history_tc, g6[tag + "_history_tc"], tag, 100)
    assert np.array_equal(r.clusters(), g6[tag + "_clusters"])
    close(r.get_covariance(), g6[tag + "_cov"], tag, 1000)


def test_tail_squash_roundtrip():
    z = np.linspace(-9, 9, 181)
    s = O.squash_tails(z)
    assert np.all(np.abs(s) < 5.0)
    assert np.allclose(s[np.abs(z) <= 4], z[np.abs(z) <= 4])
    assert np.allclose(O.unsquash_tails(s), z, atol=1e-6)


def test_schedule_and_rescale():
    assert np.allclose(O.anneal_schedule(True), [0.6, 0.36, 0.216, 0.1296, 0.07776, 0.046656, 0.0])
    assert O.anneal_schedule(False) == [0.0] and O.anneal_schedule(True, warm_start=True) == [0.0]
    w = np.ones((2, 3))
    uj = np.array([1.5, 2.0])                     # u_j >= eps0^2 * |w_j|^2 always holds (ref :249)
    out = O.rescale_for_stage(w, uj, 0.6, 0.36)
    delta = (0.36 ** 2 - 0.6 ** 2) / (1 - 0.36 ** 2) * 3.0 / uj
    a = np.sqrt((1 - 0.6 ** 2) / ((1 - 0.36 ** 2) * (1 + delta)))
    assert np.allclose(out[:, 0], 0.001 * np.floor(1000 * a))
    assert np.allclose(out * 1000, np.round(out * 1000))                       # 0.001*floor(1000 a)


# ---- synergistic branch (discourage_overlap=False): oracle vs the reference's own outputs (g8_syn.npz) -------
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_oracle_syn_matches_reference(tag):
    g = load_golden("g8_syn")
    dt = np.float32 if tag == "f32" else np.float64
    # step level: one _update_syn on the captured state
    p = "planted_%s_step_" % tag
    xt = g["planted_%s_x_tilde" % tag]
    mo_in = O.moments_syn(xt, g[p + "w_in"])
    for k in ("X_i Y_j", "cy", "ry", "rho", "X_i Z_j", "X_i^2 | Y", "TCs", "TC", "additivity", "Qi", "Si"):
        name = k.replace(" ", "_").replace("^", "p").replace("|", "g")
        assert np.allclose(mo_in[k], g[p + "in_" + name], rtol=1e-12, atol=1e-13), k
    w_out, mo_out = O.update_syn(xt, g[p + "w_in"], mo_in, eta=float(g[p + "eta"]))
    assert np.allclose(w_out, g[p + "w_out"], rtol=1e-12, atol=1e-14)
    assert abs(mo_out["TC"] - float(g[p + "out_TC"])) < 1e-11
    # end to end
    x, _ = O.gen_planted(400, 300, 5, seed=4)
    res = O.fit_syn(x, 5, seed=0, dtype=dt)
    h_ref = g["planted_%s_history_tc" % tag]
    assert len(res.history_tc) == len(h_ref)
    assert np.max(np.abs(np.asarray(res.history_tc) - h_ref)) < 1e-9
    assert np.max(np.abs(res.ws - g["planted_%s_ws" % tag])) < 1e-9
    assert np.max(np.abs(res.get_covariance() - g["planted_%s_cov" % tag])) < 1e-9
    assert np.array_equal(res.clusters(), g["planted_%s_clusters" % tag])


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("branch", ["ns", "syn"])
@pytest.mark.parametrize("gz", ["standard", "outliers"])
def test_predict_and_invert_match_reference(gz, branch, tag):
    """`predict` / `invert` (reference :431-441, g_inv :490-494) against tests/golden/g9_predict.npz (the reference's own
    outputs on big5, tests/golden/make_golden_predict.py): same arithmetic on the same inputs - exact, poles of g_inv
    (non-finite in float32, where 1 - 1e-10 rounds to 1) included."""
    g = load_golden("g9_predict")
    p = "%s_%s_%s_" % (gz, branch, tag)
    theta = (g[p + "theta_mean"], g[p + "theta_std"])
    z = g["z"].astype(DT[tag])
    with np.errstate(all="ignore"):
        inv = O.invert(z, theta, gz)
        pred = O.predict(g[p + "xz"], g[p + "y"], theta, gz)
    assert inv.dtype == g[p + "invert"].dtype
    assert np.array_equal(np.isfinite(inv), np.isfinite(g[p + "invert"]))
    ok = np.isfinite(inv)
    assert np.array_equal(inv[ok], g[p + "invert"][ok])
    close(pred, g[p + "predict"], tag)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", ["emp", "emp_missing"])
def test_empirical_gaussianize_matches_reference(name, tag):
    """gaussianize='empirical' (reference :424-426, after mean imputation :403) against the reference's own preprocess
    output (g9_predict.npz): rank transform with average ranks for ties, normal quantiles."""
    g = load_golden("g9_predict")
    x = np.array(g["emp_x_missing" if name == "emp_missing" else "emp_x"], dtype=DT[tag])
    out, theta, n_obs = O.preprocess(x, None, "empirical", -1e6 if name == "emp_missing" else None)
    assert theta is None
    assert np.array_equal(out, g["%s_%s" % (name, tag)])


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("branch", ["ns", "syn"])
@pytest.mark.parametrize("gz", ["standard", "outliers"])
def test_transform_details_matches_reference(g1, gz, branch, tag):
    """`transform(x_new, details=True)` (reference :386-395) against tests/golden/g10_transform_details.npz (the reference's own
    output, tests/golden/make_golden_transform.py): the full moments of a batch of ANOTHER row count than the fit's, preprocessed
    with the fitted theta and - as the reference does, :249 / :260 / :355 - divided by the fit's `n_samples`."""
    g = load_golden("g10_transform_details")
    p = "%s_%s_%s_" % (gz, branch, tag)
    n_fit, n_new = int(g["n_fit"]), int(g["n_new"])
    x_new = np.asarray(g1["x_raw"][n_fit:], dtype=DT[tag])
    assert len(x_new) == n_new != n_fit
    theta = (g[p + "theta_mean"], g[p + "theta_std"])
    xt, _, _ = O.preprocess(x_new, theta, gz, None)
    ws = g[p + "ws"]
    close(xt.dot(ws.T), g[p + "y_new"], tag)
    with np.errstate(all="ignore"):
        mo = O.moments_ns(xt, ws, float(g[p + "eps"]), quick=False, n_samples=n_fit) if branch == "ns" else \
            O.moments_syn(xt, ws, n_samples=n_fit)
    keys = [k[len(p + "mom_"):] for k in g.files if k.startswith(p + "mom_")]
    assert len(keys) >= 16
    have = {key_name(k): v for k, v in mo.items()}
    for k in keys:
        close(have[k], g[p + "mom_" + k], tag, scale=max(1.0, float(np.max(np.abs(g[p + "mom_" + k])))))
    # the divisor is the fit's: the batch's own row count gives another TC
    own = O.moments_ns(xt, ws, float(g[p + "eps"]), quick=False) if branch == "ns" else O.moments_syn(xt, ws)
    assert abs(float(own["TC"]) - float(g[p + "mom_TC"])) > 0.1


def test_config2_whole_fit_matches_reference():
    """BASELINE.json configs[1] END TO END against the reference itself (g12_c2_fit.npz: the float64-lifted reference's whole fit of
    RandomState(1).randn(10000, 5000), n_hidden = 32 - tests/golden/make_golden_c2fit.py): the oracle walks the same 422 iterations
    with the same 472 line-search trials, history / weights / covariance to rounding, clusters bit for bit.  This is what pins the
    CPU legs of bench.py (`cpu_fit_to_convergence`) and lets the GPU suite compare the device fit with the REFERENCE's output at
    this size instead of re-running the oracle there.  About a minute on 8 cores."""
    g = load_golden("g12_c2_fit")
    n, v, m = (int(t) for t in g["shape"])
    x = np.random.RandomState(1).randn(n, v)
    r = O.fit_ns(x, m, seed=0, dtype=np.float64)
    h, h_ref = np.asarray(r.history_tc, np.float64), g["history_tc"]
    assert len(h) == len(h_ref) == 422 and r.n_trials == int(g["trials_per_iter"].sum()) == 472
    assert r.n_trials + 10 == int(g["n_moment_calls"])
    assert np.max(np.abs(h - h_ref) / np.maximum(1.0, np.abs(h_ref))) < 1e-9
    assert np.array_equal(r.clusters(), g["clusters"])
    assert np.max(np.abs(r.ws - g["ws"])) < 1e-8 * float(np.max(np.abs(g["ws"])))
    assert np.max(np.abs(np.asarray(r.moments["TCs"]) - g["tcs"])) < 1e-8
    cov = r.get_covariance()
    assert np.max(np.abs(cov[g["cov_rows"]] - g["cov_block"])) < 1e-9 and np.max(np.abs(np.diag(cov) - g["cov_diag"])) < 1e-9
    assert abs(np.linalg.norm(cov) - float(g["cov_fro"])) < 1e-9 * float(g["cov_fro"])
