"""Parity of the HIP fit path (through the C ABI) against the CPU oracle and the golden fixtures
generated from the reference.

Tolerances (stated per SURVEY.md §8c / BASELINE.md §4):
  float64 build vs oracle-64 : step level 1e-9 relative; end to end same iteration count,
                               history/cov <= 1e-6 relative (north_star), clusters bit-exact;
  float32 build vs oracle-32 : step level 2e-4 of the array scale (the reference itself is only
                               self-consistent to ~2e-5 under summation-order changes and loses
                               more through 1/(1-rho^2)); end to end (measured with tools/f32_deviation.py:
                               iteration counts 259/263, 169/162, 225/220, final TC 1e-7..7e-7, covariance
                               3e-5..6e-4) final TC within 5e-5 relative, covariance within 2e-3 (big5) /
                               5e-4 (planted), iteration count within 6 %, clusters bit-exact on structured
                               inputs.  The reference itself moves by +-2 iterations between BLAS builds.
"""
import os

import numpy as np
import pytest

from oracle import corex_oracle as O

pytestmark = pytest.mark.gpu

DT = {"f32": np.float32, "f64": np.float64}
STEP_TOL = {"f32": 2e-4, "f64": 1e-9}


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, float(np.max(np.abs(b)))))


def make_backend(x, w, dtype):
    from linearcorex_amd.backend import HipBackend
    be = HipBackend(x.shape[0], x.shape[1], w.shape[0], dtype, 0)
    be.upload_x(x)
    be.set_ws(w)
    return be


def run_moments(be, which, eps, quick):
    be.moments_a(which)
    be.moments_b(which, eps, quick)
    be.moments_c(which)
    return be.read_state(which)


CASES = [
    # n, v, m, eps, seed
    (300, 40, 4, 0.36, 7),
    (500, 333, 5, 0.0, 8),
    (1000, 700, 16, 0.6, 9),
    (777, 1300, 20, 0.216, 10),
    (640, 512, 33, 0.1296, 11),
    (900, 260, 64, 0.0, 12),
    (400, 500, 100, 0.6, 13),
    (256, 200, 128, 0.36, 14),
    # edges: a single factor, fewer variables than a tile, sizes one past the padding granules
    (70, 3, 1, 0.0, 15),
    (64, 64, 2, 0.6, 16),
    (1000, 65, 17, 0.36, 17),
    (333, 129, 128, 0.0, 18),
    (65, 1025, 3, 0.216, 19),
    # few column tiles against a long contraction: the split of the pass goes up to 64 slots and the slots are summed
    # by the 8-threads-per-element reductions (n_samples << n_variables for X.W^T, the reverse for X^T.Y)
    (200, 8000, 6, 0.36, 20),
    (20000, 100, 5, 0.216, 21),
    (448, 6000, 30, 0.0, 22),
    # 129..256 factors: the m x m operators no longer fit the LDS of the per-variable kernels (read from L2 instead),
    # 4 wavefronts share a variable, the GEMM tiles shrink to keep 256-wide accumulators in registers
    (300, 400, 200, 0.36, 23),
    (200, 333, 256, 0.0, 24),
    (520, 1100, 129, 0.6, 25),
    # 257..1024 factors: the wide path - every contraction on gemm_wide (the factor axis tiled like any other), one thread per
    # factor in the per-variable kernels
    (220, 300, 300, 0.36, 26),
    (200, 333, 512, 0.0, 27),
    (150, 260, 600, 0.6, 28),
    (130, 1024, 1024, 0.216, 29),
]


@pytest.mark.parametrize("layout", ["auto", "panel"])
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("case", CASES)
def test_step_level(tag, case, layout, monkeypatch):
    """One evaluation, the detail moments, one update direction and one trial against the oracle, array by array.  layout "panel":
    the same on the stream-K kernel pair reading ONE panel-major copy of the shard (forced here on shapes down to 70 x 3 x 1; what
    large shards get by themselves) - same bars."""
    n, v, m, eps, seed = case
    if layout == "panel":
        if m > 256:
            pytest.skip("more than 256 factors: the wide path keeps its row-major copy")
        monkeypatch.setenv("LCX_X_LAYOUT", "panel")
        monkeypatch.setenv("LCX_PANEL_BLOCK_COLS", "192")
    dt = DT[tag]
    rng = np.random.RandomState(seed)
    x, _ = O.gen_planted(n, v, max(2, m // 2), seed=seed)
    x = O.preprocess(x.astype(dt))[0]
    w = rng.randn(m, v).astype(dt)
    w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
    w *= 3.0                                            # uj ~ 0.09: away from the trivial start
    ref = O.moments_ns(x, w, eps, quick=False)
    be = make_backend(x, w, dt)
    assert layout != "panel" or be.bytes_resident()["x_layout"].startswith("panel-major")
    assert relerr(be.download_x(), x) == 0.0
    assert relerr(be.get_ws(0), w) == 0.0
    st = run_moments(be, 0, eps, True)
    tol = STEP_TOL[tag]
    assert st[2] == 0
    assert abs(st[0] - float(ref["TC"])) <= tol * max(1.0, abs(float(ref["TC"]))) * 10
    assert abs(st[1] - float(ref["uj"].max())) <= tol
    for key in ("uj", "rho", "ry", "invrho", "rhoinvrho", "Qij", "Si", "Qi-Si^2"):
        assert relerr(be.get_moment(0, key), ref[key]) < tol * 10, key
    # detail part
    be.moments_detail(0)
    sums = be.read_sbuf(m + 3)
    assert relerr(sums[:m], ref["MI"].sum(axis=1)) < tol * 10
    assert abs(sums[m] - ref["MI"].max(axis=0).sum()) < tol * 10 * max(1.0, abs(sums[m]))
    assert abs(sums[m + 1] - ref["I(X_i ; Y)"].sum()) < tol * 10 * max(1.0, abs(sums[m + 1]))
    for key in ("MI", "X_i Z_j", "X_i^2 | Y"):
        assert relerr(be.get_moment(0, key), ref[key]) < tol * 20, key
    # update direction
    d = O.update_direction(x, w, ref, eps)
    be.update_a()
    assert relerr(be.get_moment(0, "H"), d["H"]) < tol * 10
    be.update_b(eps)
    assert relerr(be.get_moment(0, "grad"), d["grad"]) < tol * 10
    be.update_c(eps)
    be.update_d()
    assert relerr(be.get_moment(0, "sig_grad"), d["sig_grad"]) < tol * 10
    assert relerr(be.get_moment(0, "update"), d["update"]) < tol * 10
    tan = be.read_state(0)[3]
    assert abs(tan - float(d["tangent"])) <= tol * 50 * abs(float(d["tangent"]))
    # one trial, then acceptance swaps the sets
    be.make_trial(0.5)
    w_try = w + dt(0.5) * d["update"].astype(dt)
    st1 = run_moments(be, 1, eps, True)
    ref_try = O.moments_ns(x, w_try, eps, quick=True)
    if ref_try is False:
        assert st1[2] == 1
    else:
        assert st1[2] == 0
        assert abs(st1[0] - float(ref_try["TC"])) <= tol * 10 * max(1.0, abs(float(ref_try["TC"])))
    assert relerr(be.get_ws(1), w_try) < tol
    be.accept_trial()
    assert relerr(be.get_ws(0), w_try) < tol
    be.close()


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_invalid_flag_and_rescale(tag, g4):
    dt = DT[tag]
    x, w = g4["x"].astype(dt), g4["w"]
    be = make_backend(x, (w * 3.0).astype(dt), dt)
    st = run_moments(be, 0, 0.0, True)
    assert st[2] == 1 and st[1] >= 1.0 and np.isnan(st[0])                  # the `False` sentinel
    assert abs(st[1] - float(g4[tag + "_invalid_uj"].max())) < 1e-4 * st[1]
    # fixture at eps = 0.36 (generated by the reference)
    be.set_ws((w * 0.02).astype(dt))
    st = run_moments(be, 0, 0.36, False)
    tol = STEP_TOL[tag] * 10
    assert abs(st[0] - float(g4[tag + "_eps036_TC"])) < tol * max(1.0, abs(st[0]))
    for key, name in (("rho", "rho"), ("Qij", "Qij"), ("Si", "Si"), ("Qi-Si^2", "QimSip2"), ("ry", "ry"),
                      ("MI", "MI"), ("X_i Z_j", "X_i_Z_j"), ("X_i^2 | Y", "X_ip2_g_Y")):
        assert relerr(be.get_moment(0, key), g4[tag + "_eps036_" + name]) < tol, key
    # stage change rescale vs oracle
    uj = be.get_moment(0, "uj")
    ref = O.rescale_for_stage((w * 0.02).astype(dt), uj, 0.36, 0.216)
    be.rescale_ws(0.36, 0.216)
    assert relerr(be.get_ws(0), ref) < 2e-3          # floor(1000 a) may flip by one unit at a tie
    be.close()


# Iteration-count bar of the float32 end-to-end fixtures.  In float32 the TC of these fixtures (340 .. 680) is resolved to 3e-5 ..
# 6e-5, above tol = 1e-5, so `delta < tol` (:152) only fires when two consecutive TCs round to the SAME float: the count is decided
# by rounding noise, and any re-association moves it (the reference itself moves by 2 between BLAS builds; the float32
# oracle under 12 row permutations of the planted-small X: 160 .. 168 iterations; the float64 reference: 178).  Measured
# against the reference's 162 / 263 iterations (planted small / big5): "exact" 169 / 259, "exact-y" 177 / 261 (round 4).  177 is
# outside the 6 % bar the reference-shaped line search is held to - the one counter-example of the round-4 parity matrix (float64:
# identical iteration AND trial counts on every fixture), and the reason `Corex` keeps "exact" as its default; "exact-y" is held
# to 10 % here and to the same final TC / covariance bars.
F32_ITER_BAR = {"exact": 0.06, "exact-y": 0.10}


def _fit(x, m, tag, **kw):
    import os
    from linearcorex_amd import Corex
    mdl = Corex(n_hidden=m, seed=0, dtype=DT[tag], device=0, **kw).fit(x)
    # (the `ls` fixture chooses the line search through LCX_LINE_SEARCH: make sure that is the one that ran)
    assert mdl.line_search == kw.get("line_search", os.environ.get("LCX_LINE_SEARCH", "exact"))
    return mdl


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_big5_end_to_end(tag, g1, ls_both):
    ls = ls_both
    out = _fit(g1["x_raw"].astype(np.float64), 5, tag)
    h_ref = g1[tag + "_history_tc"]
    h = np.asarray(out.history["TC"], dtype=np.float64)
    assert np.array_equal(out.clusters(), g1[tag + "_clusters"])            # integer output: bit-exact
    # `moments` lists the reference dict's 20 keys (:249-287) before any of the device arrays has been read, in its insertion order
    from tests.test_host_logic_cpu import reference_moment_keys
    assert sorted(out.moments) == sorted(out.moments.keys()) == sorted(reference_moment_keys(g1, tag)) and len(out.moments) == 20
    assert list(out.moments) == list(out.moments._ORDER_DETAIL) and len(out.moments._stored()) < 20
    if tag == "f64":
        assert len(h) == len(h_ref) == 261
        assert relerr(h, h_ref) < 1e-6
        assert relerr(out.ws, g1["f64_ws"]) < 1e-6
        assert relerr(out.get_covariance(), g1["f64_cov"]) < 1e-6           # north_star tolerance
        assert relerr(out.get_covariance(rows=(7, 23)), g1["f64_cov"][7:23]) < 1e-6 and out.get_covariance(rows=slice(48, 50)).shape == (2, 50)
        with pytest.raises(ValueError):
            out.get_covariance(rows=(40, 51))
        assert out.get_covariance(rows=(9, 9)).shape == (0, 50)          # an empty block: (0, nv), as from a sharded fit
        assert relerr(out.transform(g1["x_raw"].astype(np.float64)), g1["f64_transform"]) < 1e-6
        y_res, mo_res = out.transform_fitted(details=True)          # (:392-394 for the fitted data, from what is resident: no second handle)
        assert relerr(y_res, g1["f64_transform"]) < 1e-6 and mo_res is out.moments
        assert relerr(out.moments["TCs"], g1["f64_mom_TCs"]) < 1e-6
        for key, name in (("rho", "rho"), ("MI", "MI"), ("X_i Z_j", "X_i_Z_j"), ("X_i Y_j", "X_i_Y_j"),
                          ("Qij", "Qij"), ("Si", "Si"), ("ry", "ry"), ("uj", "uj")):
            assert relerr(out.moments[key], g1["f64_mom_" + name]) < 1e-6, key
        assert abs(out.moments["additivity"] - float(g1["f64_mom_additivity"])) < 1e-6
        assert abs(out.moments["TC_no_overlap"] - float(g1["f64_mom_TC_no_overlap"])) < 1e-6
        for k, v in out.moments.items():             # every listed key materialises, with the reference's values
            from tests.test_oracle_golden import key_name
            assert relerr(v, g1["f64_mom_" + key_name(k)]) < 1e-6, k
    else:
        assert abs(len(h) - len(h_ref)) <= F32_ITER_BAR[ls] * len(h_ref)
        assert abs(float(out.tc) - float(g1["f32_tc"])) < 5e-5 * float(g1["f32_tc"])
        assert relerr(out.get_covariance(), g1["f32_cov"]) < 2e-3
        assert relerr(out.moments["TCs"], g1["f32_mom_TCs"]) < 2e-3


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_planted_small_end_to_end(tag, g2_small, ls):
    g = g2_small
    n, v, m = (int(t) for t in g["shape"])
    x, grp = O.gen_planted(n, v, m)
    out = _fit(x, m, tag)
    assert np.array_equal(out.clusters(), g[tag + "_clusters"])
    cl = out.clusters()
    assert all(len(set(cl[grp == k])) == 1 for k in range(m))
    h_ref = g[tag + "_history_tc"]
    h = np.asarray(out.history["TC"], dtype=np.float64)
    cov = out.get_covariance()
    if tag == "f64":
        assert len(h) == len(h_ref)
        assert relerr(h, h_ref) < 1e-6
        assert relerr(cov[:256, :256], g["f64_cov_block"]) < 1e-6
        assert abs(np.linalg.norm(cov) - float(g["f64_cov_fro"])) < 1e-6 * float(g["f64_cov_fro"])
        assert relerr(cov[-4:], g["f64_cov_lastrows"]) < 1e-6
    else:
        assert abs(len(h) - len(h_ref)) <= F32_ITER_BAR[ls] * len(h_ref)
        assert abs(h[-1] - h_ref[-1]) < 5e-5 * abs(h_ref[-1])
        assert relerr(cov[:256, :256], g["f32_cov_block"]) < 5e-4


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_linear_line_search_end_to_end(tag, g1, g2_small):
    """Trials without passes over X (lcx_trial_linear_*): same results as the reference-shaped mode."""
    out = _fit(g1["x_raw"].astype(np.float64), 5, tag, line_search="linear")
    assert np.array_equal(out.clusters(), g1[tag + "_clusters"])
    g = g2_small
    n, v, m = (int(t) for t in g["shape"])
    x, grp = O.gen_planted(n, v, m)
    out2 = _fit(x, m, tag, line_search="linear")
    assert np.array_equal(out2.clusters(), g[tag + "_clusters"])
    if tag == "f64":
        h, h_ref = np.asarray(out.history["TC"], np.float64), g1["f64_history_tc"]
        assert len(h) == len(h_ref) and relerr(h, h_ref) < 1e-6
        assert relerr(out.get_covariance(), g1["f64_cov"]) < 1e-6
        h2, h2_ref = np.asarray(out2.history["TC"], np.float64), g["f64_history_tc"]
        assert len(h2) == len(h2_ref) and relerr(h2, h2_ref) < 1e-6
        assert relerr(out2.get_covariance()[:256, :256], g["f64_cov_block"]) < 1e-6
        assert out2.stats["trials"] == int(g["f64_trials_per_iter"].sum())
    else:
        assert abs(float(out.tc) - float(g1["f32_tc"])) < 1e-3 * float(g1["f32_tc"])
        assert abs(float(out2.tc) - float(g["f32_tc"])) < 1e-3 * float(g["f32_tc"])
        assert relerr(out2.get_covariance()[:256, :256], g["f32_cov_block"]) < 5e-3


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_column_tiled_gemm_end_to_end(tag, g2_small, monkeypatch):
    """The whole fit with both X-streaming passes forced onto gemm_ct (the kernel large shards use:
    column-tiled waves, B through LDS, stream-K blocks) must meet the same bars as the default kernels."""
    monkeypatch.setenv("LCX_GEMM", "ct")
    g = g2_small
    n, v, m = (int(t) for t in g["shape"])
    x, grp = O.gen_planted(n, v, m)
    out = _fit(x, m, tag)
    assert "gemm_cr" in out._backend.kernel_name(0) and "gemm_ct" in out._backend.kernel_name(1)    # (the stream-K pair, panel-major copy)
    assert np.array_equal(out.clusters(), g[tag + "_clusters"])
    h_ref = g[tag + "_history_tc"]
    h = np.asarray(out.history["TC"], dtype=np.float64)
    cov = out.get_covariance()
    if tag == "f64":
        assert len(h) == len(h_ref)
        assert relerr(h, h_ref) < 1e-6
        assert relerr(cov[:256, :256], g["f64_cov_block"]) < 1e-6
    else:
        assert abs(len(h) - len(h_ref)) <= F32_ITER_BAR["exact"] * len(h_ref)
        assert abs(h[-1] - h_ref[-1]) < 5e-5 * abs(h_ref[-1])
        assert relerr(cov[:256, :256], g["f32_cov_block"]) < 5e-4


def test_f64_kernel_choice_and_fallback(g2_small, monkeypatch):
    """float64 with <= 32 factors runs its X passes on v_mfma_f64_4x4x4 (gemm_tn4); LCX_F64_MFMA=16 selects the
    16x16x4 kernel instead.  Both must reproduce the reference trajectory."""
    g = g2_small
    n, v, m = (int(t) for t in g["shape"])
    x, grp = O.gen_planted(n, v, m)
    out = _fit(x, m, "f64")
    assert "gemm_tn4_kernel" in out._backend.kernel_name(0) and "gemm_tn4_kernel" in out._backend.kernel_name(1)
    monkeypatch.setenv("LCX_F64_MFMA", "16")
    alt = _fit(x, m, "f64")
    assert "gemm_tn_kernel" in alt._backend.kernel_name(1)
    for o in (out, alt):
        h = np.asarray(o.history["TC"], dtype=np.float64)
        assert len(h) == len(g["f64_history_tc"]) and relerr(h, g["f64_history_tc"]) < 1e-6
        assert np.array_equal(o.clusters(), g["f64_clusters"])


def test_linear_trial_step_level():
    """One trial evaluated both ways on the same state must agree to rounding."""
    x, _ = O.gen_planted(600, 900, 12, seed=21)
    x = O.preprocess(x)[0]
    rng = np.random.RandomState(3)
    w = rng.randn(12, 900)
    w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
    w *= 2.0
    eps = 0.216
    be = make_backend(x, w, np.float64)
    run_moments(be, 0, eps, False)
    be.update_a(); be.update_b(eps); be.update_c(eps); be.update_d()
    for eta in (1.0, 0.5, 0.125):
        be.make_trial(eta)
        st_exact = run_moments(be, 1, eps, True)
        exact = {k: be.get_moment(1, k) for k in ("rho", "Qij", "Si", "Qi-Si^2", "uj", "ry")}
        be.trial_linear_a(eta)
        be.trial_linear_b(eps, eta)
        be.moments_c(1)
        st_lin = be.read_state(1)
        assert st_exact[2] == st_lin[2]
        if st_exact[2] == 0:
            assert abs(st_exact[0] - st_lin[0]) < 1e-10 * max(1.0, abs(st_exact[0]))
            for k, v in exact.items():
                assert relerr(be.get_moment(1, k), v) < 1e-10, k
    be.close()


def test_planted_mid_f64(g2_mid, ls):
    g = g2_mid
    n, v, m = (int(t) for t in g["shape"])
    x, grp = O.gen_planted(n, v, m)
    out = _fit(x, m, "f64")
    h_ref = g["f64_history_tc"]
    h = np.asarray(out.history["TC"], dtype=np.float64)
    assert len(h) == len(h_ref) == 242
    assert relerr(h, h_ref) < 1e-6
    assert np.array_equal(out.clusters(), g["f64_clusters"])
    cov = out.get_covariance()
    assert relerr(cov[:256, :256], g["f64_cov_block"]) < 1e-6
    assert abs(np.linalg.norm(cov) - float(g["f64_cov_fro"])) < 1e-6 * float(g["f64_cov_fro"])
    assert out.stats["trials"] == int(g["f64_trials_per_iter"].sum())
    assert out.stats["invalid_trials"] == int(g["f64_n_invalid"])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_config2_steps(tag, g3):
    """BASELINE.json config 2 shape (10^4 x 5*10^3, m=32): first iterations against the reference."""
    n, v, m = (int(t) for t in g3["shape"])
    x = O.gen_iid(n, v, seed=1, dtype=np.float64)
    out = _fit(x, m, tag, max_iter=2)
    h = np.asarray(out.history["TC"], dtype=np.float64)
    h_ref = g3[tag + "_history_tc"]
    assert len(h) == len(h_ref)
    assert relerr(h, h_ref) < (1e-7 if tag == "f64" else 2e-3)


@pytest.mark.parametrize("layout", ["auto", "panel"])
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_outliers_and_missing(tag, g5, g6, ls, layout, monkeypatch):
    """gaussianize='outliers' and missing values (-1e6 sentinel) end to end against the reference's fixtures - also with the shard
    kept as ONE panel-major copy (forced here; what large shards get by themselves), filled through two staging blocks."""
    if layout == "panel":
        monkeypatch.setenv("LCX_X_LAYOUT", "panel")
        monkeypatch.setenv("LCX_PANEL_BLOCK_COLS", "128")
    n, v, m = (int(t) for t in g5["shape"])
    x, grp = O.gen_planted(n, v, m, seed=3)
    heavy = np.arange(v) % 20 == 0
    x[:, heavy] = np.sign(x[:, heavy]) * np.abs(x[:, heavy]) ** 1.5
    out = _fit(x, m, tag, gaussianize="outliers")
    assert np.array_equal(out.clusters(), g5[tag + "_clusters"])
    tolc = 1e-6 if tag == "f64" else 5e-3
    assert relerr(out.get_covariance()[:128, :128], g5[tag + "_cov_block"]) < tolc
    if tag == "f64":
        assert len(out.history["TC"]) == len(g5["f64_history_tc"])
    # missing values (-1e6 sentinel, README recipe)
    out = _fit(g6["x_raw"], 6, tag, missing_values=-1e6, max_iter=60)
    assert np.array_equal(out.n_obs, g6[tag + "_n_obs"])
    if tag == "f64":
        assert np.array_equal(out.clusters(), g6[tag + "_clusters"])
    else:
        # unconverged (max_iter=60) fit of unstructured data in float32: argmax near-ties flip in the
        # reference itself under summation-order changes; require >= 97% agreement
        assert np.mean(out.clusters() == g6[tag + "_clusters"]) >= 0.97
    if tag == "f64":
        assert len(out.history["TC"]) == len(g6["f64_history_tc"])
        assert relerr(out.history["TC"], g6["f64_history_tc"]) < 1e-6
        assert relerr(out.get_covariance(), g6["f64_cov"]) < 1e-6


_C5_CACHE = {}


def test_config5_standin_covariance(ls):
    """BASELINE.json config 5 at full size (TCGA-OV stand-in, SURVEY.md 8d: the RPKM matrix is absent from the
    reference checkout): N=400 samples x V=20000 genes, 30 planted groups, heavy tails on 5 % of the columns,
    gaussianize='outliers', discourage_overlap (alias eliminate_synergy) on, float64.
    get_covariance() (20000 x 20000) within 1e-6 relative of the oracle, compared in row blocks;
    clusters bit-exact."""
    n, v, m = 400, 20000, 30
    x, grp = O.gen_planted(n, v, m, seed=1)
    heavy = np.arange(v) % 20 == 0
    x[:, heavy] = np.sign(x[:, heavy]) * np.abs(x[:, heavy]) ** 1.5
    if "ref" not in _C5_CACHE:
        _C5_CACHE["ref"] = O.fit_ns(x, m, seed=0, dtype=np.float64, gaussianize="outliers", max_iter=30)
    ref = _C5_CACHE["ref"]
    from linearcorex_amd import Corex
    out = Corex(n_hidden=m, seed=0, dtype=np.float64, device=0, gaussianize="outliers", eliminate_synergy=True,
                max_iter=30).fit(x)
    assert out.line_search == ls and out.stats["trials"] == ref.n_trials
    h, h_ref = np.asarray(out.history["TC"], np.float64), np.asarray(ref.history_tc)
    assert len(h) == len(h_ref)
    assert relerr(h, h_ref) < 1e-6
    assert np.array_equal(out.clusters(), ref.clusters())
    cov = out.get_covariance()
    assert cov.shape == (v, v)
    z = ref.moments["rhoinvrho"] / (1 + ref.moments["Si"])
    std = ref.theta[1]
    scale, worst = 0.0, 0.0
    for r0 in range(0, v, 2500):
        blk = np.dot(z[:, r0:r0 + 2500].T, z) / (1.0 - ref.eps ** 2)
        idx = np.arange(r0, min(v, r0 + 2500))
        blk[idx - r0, idx] = 1.0
        blk *= std[r0:r0 + 2500, np.newaxis] * std
        worst = max(worst, float(np.max(np.abs(cov[r0:r0 + 2500] - blk))))
        scale = max(scale, float(np.max(np.abs(blk))))
    assert worst / scale < 1e-6, worst / scale


def test_singular_warning_path(g4, capsys):
    """Duplicated columns: the reference hits `update_tangent >= 0` in float32 (warned=True)."""
    out = _fit(g4["dup_x"], 3, "f64", max_iter=300)
    # rank-deficient covariance: the problem is ill-conditioned, so the trajectory is only
    # reproducible to ~1e-5 for the first iterations and then amplifies last-bit differences of the
    # standardised data (device column sums vs NumPy's pairwise sums): the run towards the singular
    # solution may stop some iterations earlier or later (any change of summation order moves it: 88 iterations in
    # the reference, 70-95 here depending on how the contraction of the passes is split)
    h, h_ref = np.asarray(out.history["TC"], np.float64), g4["f64_dup_history_tc"]
    assert abs(len(h) - len(h_ref)) <= 0.3 * len(h_ref)
    assert relerr(h[:20], h_ref[:20]) < 1e-6, (relerr(h[:20], h_ref[:20]), relerr(h[:40], h_ref[:40]), len(h), h[-1], h_ref[-1])
    assert abs(h[-1] - h_ref[-1]) < 3e-2 * abs(h_ref[-1])   # TC diverges towards the singular solution
    out32 = _fit(g4["dup_x"], 3, "f32", max_iter=300)
    assert np.isfinite(out32.tc)
    # float32 on a rank-deficient covariance: TC keeps creeping towards the singular solution and the iteration at
    # which |dTC| < tol fires depends on last-bit noise (the reference itself moves by percents between BLAS builds)
    assert abs(float(out32.tc) - float(g4["f32_dup_history_tc"][-1])) < 5e-2 * abs(float(g4["f32_dup_history_tc"][-1]))


def test_pickle_roundtrip_and_warm_start(g1):
    import pickle
    x = g1["x_raw"].astype(np.float64)
    out = _fit(x, 5, "f64")
    cov = out.get_covariance()
    blob = pickle.dumps(out)
    back = pickle.loads(blob)
    assert np.array_equal(back.ws, out.ws)
    assert relerr(back.moments["rho"], out.moments["rho"]) == 0.0
    assert relerr(back.get_covariance(), cov) < 1e-12
    assert relerr(back.transform(x), out.transform(x)) < 1e-12
    # warm start: non-empty ws skips init and annealing (linearcorex.py:113-119)
    n0 = len(back.history["TC"])
    back.fit(x)
    assert len(back.history["TC"]) - n0 <= 5
    assert abs(float(back.tc) - float(out.tc)) < 1e-4


def test_generated_data_is_standardised():
    from linearcorex_amd.backend import HipBackend
    be = HipBackend(2000, 300, 4, np.float32, 0)
    be.generate_x(1, 1, 4, 0)
    x = be.download_x()
    assert np.allclose(x.mean(0), 0, atol=1e-4) and np.allclose(x.std(0), 1, atol=1e-3)
    # shard-invariance of the counter-based generator: columns 100.. of the same matrix
    be2 = HipBackend(2000, 200, 4, np.float32, 0)
    be2.generate_x(1, 1, 4, 100)
    assert np.allclose(be2.download_x(), x[:, 100:], atol=1e-6)
    # the planted group of a column as bench.py restates it (cluster purity of the convergence run): same group =>
    # correlation 1/2 (shared factor + unit noise), different groups => none
    from bench import planted_groups
    grp = planted_groups(1, 300, 4)
    c = np.corrcoef(x.astype(np.float64).T)
    same = grp[:, None] == grp[None, :]
    off = ~np.eye(300, dtype=bool)
    assert np.min(c[same & off]) > 0.4 and np.max(np.abs(c[~same])) < 0.12
    assert np.array_equal(planted_groups(1, 200, 4, col_offset=100), grp[100:])
    be.close()
    be2.close()


@pytest.mark.parametrize("shape", [(300, 6000, 8), (12000, 90, 4)])
def test_many_slots_end_to_end(shape):
    """Few samples against many variables (the regime of BASELINE.json configs[4]) and the reverse: one of the two X
    passes has only a handful of column tiles, its contraction is split into up to 64 slots.  A short float64 fit must
    follow the oracle step for step."""
    from linearcorex_amd import Corex
    n, v, m = shape
    x, _ = O.gen_planted(n, v, m, seed=31)
    ref = O.fit_ns(x, m, seed=0, dtype=np.float64, max_iter=12)
    out = Corex(n_hidden=m, seed=0, max_iter=12, dtype=np.float64, device=0).fit(x)
    geo = out._backend.geometry()
    assert max(geo["nt_split"], geo["tn_split"]) >= 12, geo
    h_ref, h_out = np.array(ref.history_tc), np.array(out.history["TC"], dtype=np.float64)
    assert len(h_ref) == len(h_out)
    assert np.max(np.abs(h_ref - h_out) / np.maximum(1.0, np.abs(h_ref))) < 1e-8
    assert np.array_equal(out.clusters(), ref.clusters())
    assert relerr(out.ws, ref.ws) < 1e-6


@pytest.mark.parametrize("tag,m", [("f64", 40), ("f32", 20)])
def test_ct_many_slots_end_to_end(tag, m):
    """Mid-size shards (4200 x 4200): the column-tiled kernel is selected with ~30 partial slots per pass (float32 from
    32 padded factors, float64 from 64), the slots are pre-reduced by the wide reductions.  A short fit must follow the
    oracle."""
    from linearcorex_amd import Corex
    n = v = 4200
    x, _ = O.gen_planted(n, v, m, seed=41)
    ref = O.fit_ns(x, m, seed=0, dtype=DT[tag], max_iter=6)
    out = Corex(n_hidden=m, seed=0, max_iter=6, dtype=DT[tag], device=0).fit(x)
    be = out._backend
    from tests.conftest import xpass_names_ok
    assert xpass_names_ok(be.kernel_name(0), be.kernel_name(1))
    geo = be.geometry()
    assert min(geo["nt_split"], geo["tn_split"]) >= 12, geo
    h_ref, h_out = np.array(ref.history_tc, np.float64), np.array(out.history["TC"], dtype=np.float64)
    assert len(h_ref) == len(h_out)
    tol = 1e-8 if tag == "f64" else 2e-3
    assert np.max(np.abs(h_ref - h_out) / np.maximum(1.0, np.abs(h_ref))) < tol
    assert np.array_equal(out.clusters(), ref.clusters())


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_init_scale_ws_against_the_reference(tag, g1):
    """`ws /= 10 * _norm(x, ws)` (reference :117, `_norm` :215-228) on its own: the weights the reference's first moment
    evaluation sees (captured by tests/golden/make_golden.py) from the same random draw."""
    xt = g1[tag + "_x_tilde"]
    np.random.seed(0)
    w0 = np.random.randn(5, xt.shape[1]).astype(DT[tag])
    be = make_backend(xt.astype(DT[tag]), w0, DT[tag])
    be.moments_a(0)
    be.init_scale_ws()
    w = be.get_ws(0)
    assert relerr(w, g1[tag + "_w_init"]) < (1e-12 if tag == "f64" else 2e-6)
    be.close()


@pytest.mark.parametrize("m", [128])        # (64 factors on the same kernels: test_merged_pass_equals_separate_passes, test_full_size_*)
def test_large_shard_kernels_inside_a_float32_fit(m, monkeypatch):
    """The instantiations BASELINE configs[2] / [3] run - gemm_ct<float, 4 | 8, ...> feeding the 64- / 128-factor float32
    epilogue, gradient and update kernels - inside a fit (7 stages x 3 iterations, the loop of reference :124-159 without
    the final factor sort, whose order is a near-tie after so few iterations), against the float32 oracle on planted data."""
    from linearcorex_amd import Corex
    from linearcorex_amd.preprocess import preprocess as pp
    monkeypatch.setenv("LCX_GEMM", "ct")
    n, v = 4096, 8192
    x, grp = O.gen_planted(n, v, m, seed=51)
    xt = pp(x.astype(np.float32), None, "standard", None)[0]
    ref = O.fit_ns_preprocessed(xt, m, seed=0, dtype=np.float32, max_iter=3, tol=0.0, finish=False)
    model = Corex(n_hidden=m, seed=0, dtype=np.float32, tol=0.0, device=0)
    be = model._attach_shard(xt, v)
    from tests.conftest import xpass_names_ok
    assert xpass_names_ok(be.kernel_name(0), be.kernel_name(1), m // 16)
    for i_eps, eps in enumerate(model._init_weights()):
        model._begin_stage(i_eps, eps)
        for k in range(3):
            model._iterate(more=k < 2)
    h_ref, h = np.asarray(ref.history_tc, np.float64), np.asarray(model.history["TC"], np.float64)
    assert len(h) == len(h_ref) == 21
    assert np.max(np.abs(h - h_ref) / np.maximum(1.0, np.abs(h_ref))) < 2e-3
    assert abs(model.stats["trials"] - ref.n_trials) <= 2
    w = be.get_ws(0)
    assert relerr(w, ref.ws) < 5e-3
    assert relerr(model.moments["rho"], ref.moments["rho"]) < 5e-3
    agree = np.mean(np.argmax(np.abs(w), axis=0) == np.argmax(np.abs(ref.ws), axis=0))
    assert agree >= 0.995, agree
    be.close()


def _iterate_n(x, m, n_iter, mode, dtype=np.float64):
    """n_iter iterations of the first annealing stage; mode: 'library' (lcx_iterate with the next iteration started early),
    'host' (levels sequenced from Python), 'mixed' (alternating: every library iteration leaves a speculation behind that the
    following host iteration has to abandon)."""
    from linearcorex_amd import Corex
    from linearcorex_amd.preprocess import preprocess as pp
    model = Corex(n_hidden=m, seed=0, dtype=dtype, tol=0.0, device=0)
    model._attach_shard(pp(x.astype(dtype), None, "standard", None)[0], x.shape[1])
    sched = model._init_weights()
    model._begin_stage(0, sched[0])
    for k in range(n_iter):
        model._in_library = mode == "library" or (mode == "mixed" and k % 2 == 0)
        model._iterate(more=True)
    model._in_library = True
    model._begin_stage(1, sched[1])          # a stage change right on top of a pending speculation
    for k in range(3):
        model._iterate(more=k < 2)
    w = model._backend.get_ws(0)
    rho = model.moments["rho"]
    stats = dict(model.stats, merged_kernel=model._backend.kernel_name(2))
    model._backend.close()
    return np.asarray(model.history["TC"], np.float64), w, rho, stats


@pytest.mark.parametrize("case", ["f64", "f32_merged_pass"])
def test_library_loop_equals_host_loop(case, monkeypatch):
    """lcx_iterate (line-search decisions in the library, next iteration's direction and first trial enqueued before the
    call returns) must walk exactly the trajectory of the host-sequenced levels - also when a speculation is abandoned by a
    host-sequenced iteration, a stage change or a readback in between.  Second case: a float32 shard on the merged pass
    (Y of the first trial already there when its evaluation starts)."""
    if case == "f64":
        x, _ = O.gen_planted(600, 900, 6, seed=61)
        dtype = np.float64
    else:
        monkeypatch.setenv("LCX_GEMM", "ct")
        x, _ = O.gen_planted(19200, 640, 6, seed=62)
        dtype = np.float32
    h_lib, w_lib, rho_lib, st_lib = _iterate_n(x, 6 if case == "f64" else 20, 14, "library", dtype)
    h_host, w_host, rho_host, st_host = _iterate_n(x, 6 if case == "f64" else 20, 14, "host", dtype)
    h_mix, w_mix, rho_mix, st_mix = _iterate_n(x, 6 if case == "f64" else 20, 14, "mixed", dtype)
    assert len(h_lib) == 17
    assert bool(st_lib["merged_kernel"]) == (case != "f64")
    assert np.array_equal(h_lib, h_host) and np.array_equal(h_lib, h_mix)
    assert np.array_equal(w_lib, w_host) and np.array_equal(w_lib, w_mix)
    assert np.array_equal(rho_lib, rho_host) and np.array_equal(rho_lib, rho_mix)
    assert st_lib["trials"] == st_host["trials"] == st_mix["trials"]
    # and against the oracle
    if case == "f64":
        ref = O.fit_ns(x, 6, seed=0, dtype=np.float64, max_iter=14)
        assert np.max(np.abs(h_lib[:14] - np.asarray(ref.history_tc[:14]))) < 1e-9


def test_host_loop_end_to_end(g1, monkeypatch):
    """LCX_HOST_LOOP=1 (what several ranks always run) on big5: same bars as the default path."""
    monkeypatch.setenv("LCX_HOST_LOOP", "1")
    out = _fit(g1["x_raw"].astype(np.float64), 5, "f64")
    assert out._in_library is False
    h, h_ref = np.asarray(out.history["TC"], np.float64), g1["f64_history_tc"]
    assert len(h) == len(h_ref) and relerr(h, h_ref) < 1e-6


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_more_than_128_factors_end_to_end(tag):
    """n_hidden above 128 (the reference takes any n_hidden, :72): a short fit with 160 factors follows the oracle, clusters
    bit-exact on planted data in float64; transform (the transposed-block path) and the synergistic branch work as well."""
    from linearcorex_amd import Corex
    n, v, m = 700, 1500, 160
    x, grp = O.gen_planted(n, v, 12, seed=71)
    ref = O.fit_ns(x, m, seed=0, dtype=DT[tag], max_iter=4, keep_x=True)
    out = Corex(n_hidden=m, seed=0, max_iter=4, dtype=DT[tag], device=0).fit(x)
    assert out._backend.geometry()["m_pad"] == 256
    h_ref, h = np.asarray(ref.history_tc, np.float64), np.asarray(out.history["TC"], np.float64)
    assert len(h) == len(h_ref)
    tol = 1e-8 if tag == "f64" else 2e-3
    assert np.max(np.abs(h - h_ref) / np.maximum(1.0, np.abs(h_ref))) < tol
    assert relerr(out.transform(x), ref.transform(ref.x_tilde)) < (1e-7 if tag == "f64" else 2e-3)
    if tag == "f64":
        assert np.array_equal(out.clusters(), ref.clusters())
        assert relerr(out.ws, ref.ws) < 1e-6
        assert relerr(out.get_covariance(), ref.get_covariance()) < 1e-6
        assert relerr(out.moments["X_i Z_j"], ref.moments["X_i Z_j"]) < 1e-6
        lin = Corex(n_hidden=m, seed=0, max_iter=4, dtype=np.float64, device=0, line_search="linear").fit(x)
        hl = np.asarray(lin.history["TC"], np.float64)
        assert len(hl) == len(h_ref) and np.max(np.abs(hl - h_ref) / np.maximum(1.0, np.abs(h_ref))) < 1e-6
        y_det, mom = out.transform(x, details=True)
        assert relerr(y_det, ref.transform(ref.x_tilde)) < 1e-7 and relerr(mom["TCs"], ref.moments["TCs"]) < 1e-6
        syn_ref = O.fit_syn(x, m, seed=0, dtype=np.float64, max_iter=5)
        syn = Corex(n_hidden=m, seed=0, max_iter=5, dtype=np.float64, device=0, discourage_overlap=False).fit(x)
        hs, hs_ref = np.asarray(syn.history["TC"], np.float64), np.asarray(syn_ref.history_tc)
        assert len(hs) == len(hs_ref) and np.max(np.abs(hs - hs_ref) / np.maximum(1.0, np.abs(hs_ref))) < 1e-8


# (520 factors float32 and 300 factors float64 end to end - 8 + 7 s on the untuned wide path - left the suite in round 6; 300 / 512 / 600 / 1024
# factors in both precisions stay in test_step_level)
@pytest.mark.parametrize("tag,m", [("f32", 300)])
def test_more_than_256_factors_end_to_end(tag, m):
    """n_hidden above 256 (the reference takes any n_hidden, :72) on the wide path (m_pad 512 / 1024): a short fit follows the
    oracle, clusters bit-exact on planted data in float64; transform, predict, get_covariance, the linear trial mode and the
    synergistic branch work as well."""
    from linearcorex_amd import Corex
    n, v = 500, 900
    x, grp = O.gen_planted(n, v, 12, seed=72)
    ref = O.fit_ns(x, m, seed=0, dtype=DT[tag], max_iter=3, keep_x=True)
    out = Corex(n_hidden=m, seed=0, max_iter=3, dtype=DT[tag], device=0).fit(x)
    assert out._backend.geometry()["m_pad"] == (512 if m <= 512 else 1024)
    assert "gemm_wide_kernel" in out._backend.kernel_name(0)
    h_ref, h = np.asarray(ref.history_tc, np.float64), np.asarray(out.history["TC"], np.float64)
    assert len(h) == len(h_ref)
    tol = 1e-8 if tag == "f64" else 2e-3
    assert np.max(np.abs(h - h_ref) / np.maximum(1.0, np.abs(h_ref))) < tol
    assert relerr(out.transform(x), ref.transform(ref.x_tilde)) < (1e-7 if tag == "f64" else 2e-3)
    if tag == "f64":
        assert np.array_equal(out.clusters(), ref.clusters())
        assert relerr(out.ws, ref.ws) < 1e-6
        assert relerr(out.get_covariance(), ref.get_covariance()) < 1e-6
        assert relerr(out.moments["X_i Z_j"], ref.moments["X_i Z_j"]) < 1e-6
        y = ref.transform(ref.x_tilde)[:40]
        assert relerr(out.predict(y), O.predict(ref.moments["X_i Z_j"], y, ref.theta)) < 1e-6
        if m == 300:
            lin = Corex(n_hidden=m, seed=0, max_iter=3, dtype=np.float64, device=0, line_search="linear").fit(x)
            hl = np.asarray(lin.history["TC"], np.float64)
            assert len(hl) == len(h_ref) and np.max(np.abs(hl - h_ref) / np.maximum(1.0, np.abs(h_ref))) < 1e-6
            syn_ref = O.fit_syn(x, m, seed=0, dtype=np.float64, max_iter=4)
            syn = Corex(n_hidden=m, seed=0, max_iter=4, dtype=np.float64, device=0, discourage_overlap=False).fit(x)
            hs, hs_ref = np.asarray(syn.history["TC"], np.float64), np.asarray(syn_ref.history_tc)
            assert len(hs) == len(hs_ref) and np.max(np.abs(hs - hs_ref) / np.maximum(1.0, np.abs(hs_ref))) < 1e-8


def test_more_than_1024_factors_is_refused():
    from linearcorex_amd import Corex
    from linearcorex_amd._abi import LcxError
    with pytest.raises(LcxError, match="n_hidden > 1024"):
        Corex(n_hidden=1025, seed=0, device=0).fit(np.random.RandomState(0).randn(50, 300))


@pytest.mark.parametrize("m", [24, 64])
def test_merged_pass_equals_separate_passes(m, monkeypatch):
    """float32 shards on gemm_ct with <= 64 padded factors run X.grad^T (:210) and the first trial's X.(ws + update)^T (:321) as
    ONE pass over X with twice the columns.  Same arithmetic per element up to the summation split of the wider kernel: the
    fit must follow the two-pass path (LCX_MERGED_PASS=0) to float32 rounding, and the float32 oracle within the usual bar."""
    from linearcorex_amd import Corex
    monkeypatch.setenv("LCX_GEMM", "ct")
    x, _ = O.gen_planted(19200, 1280, 8, seed=81)      # 75 super tiles of samples: the merged pass splits into <= 8 slots
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LCX_MERGED_PASS", flag)
        out = Corex(n_hidden=m, seed=0, max_iter=6, dtype=np.float32, device=0).fit(x)
        assert bool(out._backend.kernel_name(2)) == (flag == "1")
        runs[flag] = (np.asarray(out.history["TC"], np.float64), out.stats["trials"], out.ws.copy(), out.clusters())
        out._backend.close()
    h1, h0 = runs["1"][0], runs["0"][0]
    assert len(h1) == len(h0) == 42 and runs["1"][1] == runs["0"][1]
    assert np.max(np.abs(h1 - h0) / np.maximum(1.0, np.abs(h0))) < 5e-5
    assert relerr(runs["1"][2], runs["0"][2]) < 1e-3
    ref = O.fit_ns(x, m, seed=0, dtype=np.float32, max_iter=6)
    h_ref = np.asarray(ref.history_tc, np.float64)
    assert len(h_ref) == len(h1) and np.max(np.abs(h1 - h_ref) / np.maximum(1.0, np.abs(h_ref))) < 2e-3


@pytest.mark.parametrize("m", [64, 128])
def test_linear_mode_float32_large_shards(m, monkeypatch):
    """The linear trial mode (trials cost no pass over X: Y and X^T.Y of ws + eta*update follow from the direction's passes,
    re-anchored on an exact evaluation every 16 iterations and at every stage) on the kernels BASELINE configs[2] / [3] run:
    float32, gemm_ct, n_hidden 64 / 128, 4096 x 8192, 7 stages x 20 iterations.  Same mathematics as the exact mode, different
    rounding: the TC history must stay within the float32 end-to-end bar of the exact mode (5e-5 relative; measured 3e-7..8e-7),
    trial counts within 2, and the weights must describe the same solution."""
    from linearcorex_amd import Corex
    from linearcorex_amd.preprocess import preprocess as pp
    monkeypatch.setenv("LCX_GEMM", "ct")
    n, v, iters = 4096, 8192, 20
    x, _ = O.gen_planted(n, v, m, seed=51)
    xt = pp(x.astype(np.float32), None, "standard", None)[0]
    runs = {}
    for mode in ("exact", "linear"):
        model = Corex(n_hidden=m, seed=0, dtype=np.float32, tol=0.0, device=0, line_search=mode)
        be = model._attach_shard(xt, v)
        for i_eps, eps in enumerate(model._init_weights()):
            model._begin_stage(i_eps, eps)
            for k in range(iters):
                model._iterate(more=k + 1 < iters)
        runs[mode] = (np.asarray(model.history["TC"], np.float64), be.get_ws(0), dict(model.stats))
        be.close()
    (h0, w0, s0), (h1, w1, s1) = runs["exact"], runs["linear"]
    assert len(h0) == len(h1) == 7 * iters
    assert s1.get("refreshes", 0) >= 7                     # the 16-iteration re-anchor fired in every stage
    dev = np.max(np.abs(h1 - h0) / np.maximum(1.0, np.abs(h0)))
    assert dev < 5e-5, dev
    assert abs(s1["trials"] - s0["trials"]) <= 2, (s0["trials"], s1["trials"])
    assert relerr(w1, w0) < 2e-2
    agree = np.mean(np.argmax(np.abs(w1), axis=0) == np.argmax(np.abs(w0), axis=0))
    assert agree >= 0.995, agree
    print("linear vs exact float32 m=%d: max rel TC deviation %.2e, trials %d vs %d, ws %.2e, clusters agree %.4f"
          % (m, dev, s1["trials"], s0["trials"], relerr(w1, w0), agree))


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_single_copy_mode_end_to_end(tag, monkeypatch):
    """One resident copy of the shard (LCX_SINGLE_COPY=1; automatic when two would not fit): X.B^T is read from the row-major X
    itself by gemm_cr instead of from the transposed copy.  Same fit as the two-copy path to rounding, and vs the oracle at
    the usual bars; the handle owns half the X bytes."""
    from linearcorex_amd import Corex
    dt = DT[tag]
    x = O.gen_planted(500, 2000, 8, seed=1)[0]
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LCX_SINGLE_COPY", flag)
        out = Corex(n_hidden=8, seed=0, dtype=dt, device=0, max_iter=12, tol=0.0).fit(x)      # tol 0: the same 7 x 12 iterations everywhere
        be = out._backend
        assert ("gemm_cr_kernel" in be.kernel_name(0)) == (flag == "1")
        br = be.bytes_resident()
        runs[flag] = (np.asarray(out.history["TC"], np.float64), out.ws.copy(), out.clusters(), br["x"],
                      out.get_covariance(), out.transform(x))
        be.close()
    (h1, w1, c1, b1, cov1, y1), (h0, w0, c0, b0, cov0, y0) = runs["1"], runs["0"]
    assert b1 * 2 == b0
    assert len(h1) == len(h0)
    tol = 1e-9 if tag == "f64" else 5e-4
    assert np.max(np.abs(h1 - h0) / np.maximum(1.0, np.abs(h0))) < tol
    assert relerr(w1, w0) < tol * 10 and relerr(cov1, cov0) < tol * 10 and relerr(y1, y0) < tol * 10
    ref = O.fit_ns(x, 8, seed=0, dtype=dt, max_iter=12, tol=0.0)
    hr = np.asarray(ref.history_tc, np.float64)
    assert len(hr) == len(h1) == 84
    assert np.max(np.abs(h1 - hr) / np.maximum(1.0, np.abs(hr))) < (1e-8 if tag == "f64" else 2e-3)
    if tag == "f64":
        assert np.array_equal(c1, ref.clusters())


def test_single_copy_merged_pass_and_empirical(monkeypatch):
    """single-copy mode on the paths that used the transposed copy for something else: the merged pass (gemm_cr with 2 m_pad
    columns) and gaussianize='empirical' (a transposed copy only for the duration of the sort)"""
    from linearcorex_amd import Corex
    x, _ = O.gen_planted(19200, 1280, 8, seed=81)
    runs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LCX_SINGLE_COPY", flag)
        out = Corex(n_hidden=24, seed=0, max_iter=4, dtype=np.float32, device=0).fit(x)
        assert ("gemm_cr_kernel" in out._backend.kernel_name(2)) == (flag == "1")
        runs[flag] = (np.asarray(out.history["TC"], np.float64), out.stats["trials"])
        out._backend.close()
    assert len(runs["1"][0]) == len(runs["0"][0]) == 28 and runs["1"][1] == runs["0"][1]
    assert np.max(np.abs(runs["1"][0] - runs["0"][0]) / np.maximum(1.0, np.abs(runs["0"][0]))) < 5e-5
    monkeypatch.setenv("LCX_SINGLE_COPY", "1")
    xe = np.random.RandomState(3).lognormal(size=(700, 90))
    out = Corex(n_hidden=3, seed=0, dtype=np.float64, device=0, max_iter=5, gaussianize="empirical").fit(xe)
    xr = O.preprocess(xe.copy(), None, "empirical")[0]
    orc = O.fit_ns(xr, 3, seed=0, dtype=np.float64, max_iter=5, gaussianize="none")
    assert np.max(np.abs(np.asarray(out.history["TC"], np.float64) - np.asarray(orc.history_tc))) < 1e-9
    out._backend.close()


@pytest.mark.parametrize("tag", ["f32", "f64", "f32_merged"])
def test_later_trials_by_linearity(tag, monkeypatch):
    """line_search='exact-y' (lcx_set_trial_reuse): the back-tracking trials after the first one of an iteration take
    X.w_update^T = Y + eta X.update^T from the iteration's own exact products and make one pass over X instead of two.  An exact
    re-association: the trajectory must equal the reference-shaped one to rounding - float64 1e-10 on the TC history, float32 on
    the large-shard kernels (gemm_ct, 4096 x 8192, 128 factors) within 5e-5 - with the same number of trials in every iteration
    that still moves TC (on a converged plateau the Wolfe test compares rounding noise and either run may halve eta a few more
    times), and it must really save passes: every trial after the first skips its X.B^T launch."""
    from linearcorex_amd import Corex
    from linearcorex_amd.preprocess import preprocess as pp
    if tag == "f64":
        x, _ = O.gen_planted(700, 900, 6, seed=91)
        m, iters, dt = 6, 25, np.float64
    elif tag == "f32":
        monkeypatch.setenv("LCX_GEMM", "ct")
        x, _ = O.gen_planted(4096, 8192, 128, seed=51)
        m, iters, dt = 128, 12, np.float32
    else:                                   # the merged pass X.[grad | ws + update]^T in front of the trials
        monkeypatch.setenv("LCX_GEMM", "ct")
        x, _ = O.gen_planted(19200, 1280, 8, seed=81)
        m, iters, dt = 24, 8, np.float32
    xt = pp(x.astype(dt), None, "standard", None)[0]
    runs = {}
    for mode in ("exact", "exact-y"):
        model = Corex(n_hidden=m, seed=0, dtype=dt, tol=0.0, device=0, line_search=mode)
        be = model._attach_shard(xt, x.shape[1])
        assert bool(be.kernel_name(2)) == (tag == "f32_merged")
        be.timing_enable(True)
        per = []
        for i_eps, eps in enumerate(model._init_weights()):
            model._begin_stage(i_eps, eps)
            for k in range(iters):
                t0 = model.stats["trials"]
                model._iterate(more=k + 1 < iters)
                per.append(model.stats["trials"] - t0)
        passes = be.timing_passes_by_kind()
        runs[mode] = (np.asarray(model.history["TC"], np.float64), be.get_ws(0), np.asarray(per), passes)
        be.close()
    (h0, w0, t0, p0), (h1, w1, t1, p1) = runs["exact"], runs["exact-y"]
    assert len(h0) == len(h1) == 7 * iters
    tol = 1e-10 if tag == "f64" else 5e-5
    assert np.max(np.abs(h1 - h0) / np.maximum(1.0, np.abs(h0))) < tol
    assert relerr(w1, w0) < (1e-8 if tag == "f64" else 5e-3)
    prev = np.concatenate([[h0[0] - 1.0], h0[:-1]])
    moving = np.abs(h0 - prev) > (1e-9 if tag == "f64" else 1e-5) * np.abs(h0)         # iterations that still move TC
    assert moving.sum() > 3 * iters and t0[moving].sum() > moving.sum() + 3             # ... and back-track now and then
    assert np.array_equal(t0[moving], t1[moving])
    # every trial after the first one of its iteration skipped its X.B^T pass; the X^T.Y passes are all there
    nt1 = p1["gemm_nt"] + p1["gemm_nt2"]
    assert nt1 <= 7 * iters * 2 + 20, (nt1, p1)
    assert p0["gemm_nt"] + p0["gemm_nt2"] - nt1 >= (t1 - 1).sum() - 14
    if tag == "f64":
        ref = O.fit_ns_preprocessed(xt, m, seed=0, dtype=dt, max_iter=iters, tol=0.0, finish=False)
        assert np.max(np.abs(h1 - np.asarray(ref.history_tc)) / np.maximum(1.0, np.abs(h1))) < 1e-8


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_later_trials_by_linearity_whole_fit(tag, g1):
    """line_search='exact-y' through `fit()` itself (stage changes, convergence test, final detail moments and factor sort) on
    big5: the result of the reference-shaped fit, to rounding."""
    from linearcorex_amd import Corex
    dt = DT[tag]
    a = Corex(n_hidden=5, seed=0, dtype=dt, device=0).fit(g1["x_raw"])
    b = Corex(n_hidden=5, seed=0, dtype=dt, device=0, line_search="exact-y").fit(g1["x_raw"])
    ha, hb = np.asarray(a.history["TC"], np.float64), np.asarray(b.history["TC"], np.float64)
    if tag == "f64":
        assert len(ha) == len(hb) == len(g1["f64_history_tc"])
        assert np.max(np.abs(hb - ha) / np.maximum(1.0, np.abs(ha))) < 1e-9
        assert np.array_equal(a.clusters(), b.clusters()) and relerr(b.get_covariance(), a.get_covariance()) < 1e-8
        assert relerr(b.get_covariance(), g1["f64_cov"]) < 1e-6
    else:
        assert abs(len(ha) - len(hb)) <= 0.06 * len(ha)
        assert abs(float(b.tc) - float(a.tc)) < 5e-5 * abs(float(a.tc))
        assert np.array_equal(a.clusters(), b.clusters())
    assert b.stats["trials"] > len(hb)
    a._backend.close()
    b._backend.close()


# ------------------------------------------------------------------------------------------------------------------------------
# round 4: ONE panel-major resident copy of the shard for large shards (include/lcx.h, lcx_x_layout; gemm_kernels.hpp, PanelW)
# ------------------------------------------------------------------------------------------------------------------------------
# (round 6: one precision per shape - the other halves, (1500, 3000, 8) f64, (1501, 3001, 40) f32, (1300, 2500, 100) f32, made room for the
# bench-job tests; LCX_MORE_LAYOUT_CASES=1 brings them back)
_PANEL_CASES = [((1500, 3000, 8), "f32"), ((1501, 3001, 40), "f64"), ((2048, 4096, 64), "f32"), ((1300, 2500, 100), "f64"), ((900, 2100, 200), "f32")]
if os.environ.get("LCX_MORE_LAYOUT_CASES"):
    _PANEL_CASES += [((1500, 3000, 8), "f64"), ((1501, 3001, 40), "f32"), ((1300, 2500, 100), "f32")]


@pytest.mark.parametrize("shape,tag", _PANEL_CASES)
def test_panel_layout_matches_row_major(tag, shape, monkeypatch):
    """The panel-major copy (both X passes on the stream-K kernels from the same bytes) against the row-major + transposed layout on
    the same kernels: X.B^T contracts in another order (rounding), X^T.Y in the same one; same fit to rounding, both at the usual
    bars against the oracle; the handle owns HALF the X bytes; the resident matrix reads back identical (ragged shapes, several
    staging blocks).  Factor counts across the tile shapes of CtShape: 16 / 64 padded columns (4 row tiles per wave, float64 2), 128
    (float32: 8 waves per block), 256 (2 / 1 row tiles: 8-byte pieces in float64)."""
    from linearcorex_amd import Corex
    dt = DT[tag]
    n, v, m = shape
    if tag == "f64" and m == 64:
        pytest.skip("float32 shape")
    x = O.gen_planted(n, v, min(m, 8), seed=1)[0]
    monkeypatch.setenv("LCX_PANEL_BLOCK_COLS", "832")            # 13 panels of float32 per block: 4 .. 10 blocks, the last one ragged
    runs = {}
    for lay in ("panel", "rows"):
        monkeypatch.setenv("LCX_X_LAYOUT", lay)
        monkeypatch.setenv("LCX_GEMM", "ct")
        out = Corex(n_hidden=m, seed=0, dtype=dt, device=0, max_iter=6, tol=0.0).fit(x)
        be = out._backend
        br = be.bytes_resident()
        assert br["x_layout"].startswith("panel-major" if lay == "panel" else "row-major + transposed")
        from tests.conftest import xpass_names_ok
        if lay == "panel":
            assert xpass_names_ok(be.kernel_name(0), be.kernel_name(1))
        else:
            assert "gemm_ct_kernel" in be.kernel_name(0) and "gemm_ct_kernel" in be.kernel_name(1) and be.kernel_name(1).endswith("true, false>")
        runs[lay] = (np.asarray(out.history["TC"], np.float64), out.ws.copy(), out.clusters(), br["x"], be.download_x(), out.transform(x),
                     out.stats["trials"])
        be.close()
    (h1, w1, c1, b1, x1, y1, t1), (h0, w0, c0, b0, x0, y0, t0) = runs["panel"], runs["rows"]
    assert b1 * 2 == b0 and len(h1) == len(h0) == 42
    # the preprocessed shard itself: per-column statistics are summed per staging block in the panel path (another split of the rows)
    assert relerr(x1, x0) < (1e-14 if tag == "f64" else 2e-6)
    tol = 1e-9 if tag == "f64" else 5e-4
    assert np.max(np.abs(h1 - h0) / np.maximum(1.0, np.abs(h0))) < tol
    assert relerr(w1, w0) < tol * 10 and relerr(y1, y0) < tol * 10
    if tag == "f64":
        assert t1 == t0 and np.array_equal(c1, c0)
    ref = O.fit_ns(x, m, seed=0, dtype=dt, max_iter=6, tol=0.0)
    hr = np.asarray(ref.history_tc, np.float64)
    assert np.max(np.abs(h1 - hr) / np.maximum(1.0, np.abs(hr))) < (1e-8 if tag == "f64" else 2e-3)


@pytest.mark.parametrize("gz,missing", [("outliers", False), ("standard", True), ("empirical", True), ("none", False)])
def test_panel_layout_preprocess_and_generate(gz, missing, monkeypatch):
    """Every preprocessing kind of :397-429 through the staged blocks of the panel layout (per-column work: blocks of columns are
    independent) against the row-major path: theta, n_obs and the resident matrix; and the on-device generator block by block."""
    from linearcorex_amd.backend import HipBackend
    rng = np.random.RandomState(3)
    n, v = 1300, 2100
    x = rng.randn(n, v) * (1 + rng.rand(v)) + rng.randn(v)
    x[:, ::7] = np.sign(x[:, ::7]) * np.abs(x[:, ::7]) ** 1.5
    x[:, 5] = np.round(x[:, 5])                      # ties for the rank transform
    if missing:
        x[rng.rand(n, v) < 0.03] = -1e6
    monkeypatch.setenv("LCX_GEMM", "ct")
    monkeypatch.setenv("LCX_PANEL_BLOCK_COLS", "512")
    got = {}
    for lay in ("panel", "rows"):
        monkeypatch.setenv("LCX_X_LAYOUT", lay)
        be = HipBackend(n, v, 8, np.float64, 0)
        theta, n_obs, mx = be.upload_preprocess(x, gz, -1e6 if missing else None, None)
        got[lay] = (theta, n_obs, mx, be.download_x())
        # a second batch with the fitted theta (transform-time preprocessing of a handle of its own)
        if gz in ("standard", "outliers"):
            be.upload_preprocess(x[::-1].copy(), gz, -1e6 if missing else None, theta)
            got[lay] += (be.download_x(),)
        be.generate_x(7, 1, 5, 1000)
        got[lay] += (be.download_x(),)
        be.close()
    a, b = got["panel"], got["rows"]
    if gz in ("standard", "outliers"):
        assert relerr(a[0][0], b[0][0]) < 1e-13 and relerr(a[0][1], b[0][1]) < 1e-13
    assert np.array_equal(a[1], b[1]) and abs(a[2] - b[2]) <= 1e-12 * max(1.0, abs(b[2]))
    for u, w in zip(a[3:], b[3:]):
        assert u.shape == w.shape and relerr(u, w) < 1e-12
    ref = O.preprocess(x.astype(np.float64), None, gz, -1e6 if missing else None)[0]
    assert relerr(a[3], ref) < 1e-10


# ------------------------------------------------------------------------------------------------------------------------------
# round 4: the X passes of a float32 panel shard on the bf16 matrix pipe (include/lcx.h, lcx_set_f32_gemm; gemm_split_kernels.hpp)
# ------------------------------------------------------------------------------------------------------------------------------
# (round 5: one shape per tile width - 32 / 64 / 128 padded factors, the first two with their merged pass; the opt-in is a rider
# and the full six-shape matrix of round 4, incl. (2048, 4096, 64), (1300, 2500, 100), (20000, 1500, 40), cost a minute of the suite)
# (round 6: 64 padded factors with the merged pass, and 128; the 32-factor shape (1500, 3000, 20) left the suite with the other rider repeats)
@pytest.mark.parametrize("shape", [(1501, 3001, 40), (2048, 1100, 128)])
def test_split_gemm_matches_mfma(shape, monkeypatch):
    """f32_gemm="split" (every operand split exactly into three bf16 numbers, 6 partial products, float32 accumulation) against
    f32_gemm="mfma" (float32 MFMA) on the same panel-major shard: same fit to float32 rounding, both at the float32 bar against the
    oracle, and - the precision claim - the split moments are as close to a FLOAT64 fit as the float32-MFMA moments are (within 2 x).
    Factor counts over the three tile widths (32 / 64 / 128 padded columns), ragged sizes, contractions that are an odd number of
    32-element groups; the 20 000-row shape also runs the merged pass (twice the columns: 128)."""
    from linearcorex_amd import Corex
    n, v, m = shape
    x = O.gen_planted(n, v, min(m, 8), seed=1)[0]
    monkeypatch.setenv("LCX_X_LAYOUT", "panel")
    monkeypatch.setenv("LCX_GEMM", "ct")
    runs = {}
    merged = False
    for gemm in ("split", "mfma"):
        out = Corex(n_hidden=m, seed=0, dtype=np.float32, device=0, max_iter=6, tol=0.0, f32_gemm=gemm).fit(x)
        be = out._backend
        assert out.f32_gemm == gemm == be.f32_gemm()
        assert ("gemm_split_kernel" in be.kernel_name(0)) == (gemm == "split") == ("gemm_split_kernel" in be.kernel_name(1))
        if be.kernel_name(2):                                    # the merged pass, where the shard runs one
            assert ("gemm_split_kernel" in be.kernel_name(2)) == (gemm == "split")
            merged = True
        runs[gemm] = (np.asarray(out.history["TC"], np.float64), out.ws.copy(), out.transform(x), out.moments["rho"].copy(),
                      out.moments["uj"].copy(), out.clusters())
        be.close()
    (h1, w1, y1, r1, u1, c1), (h0, w0, y0, r0, u0, c0) = runs["split"], runs["mfma"]
    assert merged == (n >= 20000)                                # the shapes here whose merged launch needs <= 8 slots
    assert len(h1) == len(h0) == 42
    assert np.max(np.abs(h1 - h0) / np.maximum(1.0, np.abs(h0))) < 5e-4
    # the rows of ws are sorted by TCs at the end of a fit (:160-163): factors whose TCs tie to rounding may swap places
    cn = (w1 / np.linalg.norm(w1, axis=1, keepdims=True)).dot((w0 / np.linalg.norm(w0, axis=1, keepdims=True)).T)
    perm = np.argmax(np.abs(cn), axis=1)
    assert sorted(perm) == list(range(m)) and np.sum(perm != np.arange(m)) <= m // 2
    assert relerr(w1, w0[perm]) < 5e-3 and relerr(y1, y0[:, perm]) < 5e-3
    r0, u0 = r0[perm], u0[perm]
    ref = O.fit_ns(x, m, seed=0, dtype=np.float32, max_iter=6, tol=0.0)
    hr = np.asarray(ref.history_tc, np.float64)
    assert np.max(np.abs(h1 - hr) / np.maximum(1.0, np.abs(hr))) < 2e-3
    # against float64: the split fit is no further from it than the float32-MFMA fit (x 2) - or both are at the noise floor
    r64 = O.fit_ns(x, m, seed=0, dtype=np.float64, max_iter=6, tol=0.0)
    h64 = np.asarray(r64.history_tc, np.float64)
    e1 = np.max(np.abs(h1 - h64) / np.maximum(1.0, np.abs(h64)))
    e0 = np.max(np.abs(h0 - h64) / np.maximum(1.0, np.abs(h64)))
    assert e1 < 2.0 * e0 + 2e-6, (e1, e0)
    c64 = (w1 / np.linalg.norm(w1, axis=1, keepdims=True)).dot((r64.ws / np.linalg.norm(r64.ws, axis=1, keepdims=True)).T)
    p64 = np.argmax(np.abs(c64), axis=1)
    assert sorted(p64) == list(range(m))
    for a1, a0, key in ((r1, r0, "rho"), (u1, u0, "uj")):
        d1, d0 = relerr(a1, r64.moments[key][p64]), relerr(a0, r64.moments[key][p64])
        assert d1 < 2.0 * d0 + 2e-6, (key, d1, d0)


def test_split_gemm_one_pass_error_vs_float64(monkeypatch):
    """One evaluation of the moments (both X passes) of the same W in both float32 modes against the float64 oracle, element by
    element: Y = X.W^T, rho (from X^T.Y) and uj.  The split contraction's error is within 2 x that of the float32 MFMA."""
    from linearcorex_amd.backend import HipBackend
    monkeypatch.setenv("LCX_X_LAYOUT", "panel")
    monkeypatch.setenv("LCX_GEMM", "ct")
    n, v, m = 6000, 9000, 64
    x = O.gen_planted(n, v, 8, seed=2)[0].astype(np.float32)
    x = (x - x.mean(0)) / x.std(0)
    rng = np.random.RandomState(5)
    w = rng.randn(m, v).astype(np.float32)
    w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
    w *= np.float32(3.0)
    ref = O.moments_ns(x.astype(np.float64), w.astype(np.float64), 0.36, quick=True)
    y64 = x.astype(np.float64).dot(w.astype(np.float64).T)
    be = HipBackend(n, v, m, np.float32, 0)
    be.upload_x(x)
    be.set_ws(w)
    err = {}
    for gemm in ("mfma", "split", "mfma"):                       # switchable between launches, and back
        assert be.set_f32_gemm(gemm) == gemm
        be.moments_a(0); be.moments_b(0, 0.36, 1); be.moments_c(0)
        st = be.read_state(0)
        assert st[2] == 0
        err[gemm] = (relerr(be.get_moment(0, "Y")[:n], y64), relerr(be.get_moment(0, "rho"), ref["rho"]), relerr(be.get_moment(0, "uj"), ref["uj"]),
                     abs(st[0] - float(ref["TC"])) / abs(float(ref["TC"])))
    be.close()
    print("one pass vs float64 (Y, rho, uj, TC):", {k: ["%.2e" % e for e in v_] for k, v_ in err.items()})
    for k in range(4):
        assert err["split"][k] < 2.0 * err["mfma"][k] + 1e-6, (k, err)
        assert err["split"][k] < 1e-4


@pytest.mark.parametrize("tag,m", [("f64", 40), ("f32", 8), ("f32", 200)])
def test_split_gemm_is_refused_where_it_does_not_apply(tag, m, monkeypatch):
    """float64, 16 padded factors and more than 128: the call succeeds, the mode stays "mfma", the fit is the float32 / float64 one."""
    from linearcorex_amd import Corex
    monkeypatch.setenv("LCX_X_LAYOUT", "panel")
    monkeypatch.setenv("LCX_GEMM", "ct")
    x = O.gen_planted(900, 2100, 8, seed=1)[0]
    out = Corex(n_hidden=m, seed=0, dtype=DT[tag], device=0, max_iter=2, tol=0.0, f32_gemm="split").fit(x)
    assert out.f32_gemm == "mfma" and "split" not in out._backend.kernel_name(0)
    out._backend.close()
    monkeypatch.setenv("LCX_X_LAYOUT", "rows")                   # and a float32 shard that is not in the panel layout
    out = Corex(n_hidden=40, seed=0, dtype=np.float32, device=0, max_iter=2, tol=0.0, f32_gemm="split").fit(x)
    assert out.f32_gemm == "mfma"
    out._backend.close()


def test_split_gemm_env_default(monkeypatch):
    """LCX_F32_GEMM=split makes it the library's default for the handles that support it; an explicit f32_gemm="mfma" wins."""
    from linearcorex_amd import Corex
    monkeypatch.setenv("LCX_X_LAYOUT", "panel")
    monkeypatch.setenv("LCX_GEMM", "ct")
    monkeypatch.setenv("LCX_F32_GEMM", "split")
    x = O.gen_planted(900, 2100, 8, seed=1)[0]
    a = Corex(n_hidden=40, seed=0, dtype=np.float32, device=0, max_iter=2, tol=0.0).fit(x)
    b = Corex(n_hidden=40, seed=0, dtype=np.float32, device=0, max_iter=2, tol=0.0, f32_gemm="mfma").fit(x)
    assert a.f32_gemm == "split" and b.f32_gemm == "mfma"
    assert "gemm_split_kernel" in a._backend.kernel_name(0) and "gemm_split_kernel" not in b._backend.kernel_name(0)
    a._backend.close(); b._backend.close()


@pytest.mark.parametrize("shape", [(2048, 8200, 64), (1100, 5003, 100), (3000, 1037, 128), (20000, 1500, 40)])
def test_per_variable_kernels_on_the_matrix_pipe(shape, monkeypatch):
    """moments_epilogue / grad with the m x m operator product on MFMA (a wave per 16 variables, float32, 64 / 128 padded factors)
    against the thread-per-(variable, factor) forms (LCX_PV_MFMA=0) on the same handle: every array they write - D, rho, rhoinvrho,
    Qij, Si, Qi-Si^2, grad and the scalars built from their per-block partials (TC, the Bj of the direction, update_tangent) - to
    float32 rounding, on variable counts that are not multiples of 16, with and without the merged pass; and a whole exact-mode and
    linear-mode fit (the linear trials feed the epilogue from D + eta D(update)) lands on the same history."""
    from linearcorex_amd import Corex
    from linearcorex_amd.backend import HipBackend
    n, v, m = shape
    monkeypatch.setenv("LCX_X_LAYOUT", "panel")
    monkeypatch.setenv("LCX_GEMM", "ct")
    x = O.gen_planted(n, v, 8, seed=3)[0].astype(np.float32)
    x = (x - x.mean(0)) / x.std(0)
    rng = np.random.RandomState(6)
    w = rng.randn(m, v).astype(np.float32)
    w /= (10.0 * O.norm(x, w, 0))[:, np.newaxis]
    w *= np.float32(3.0)
    be = HipBackend(n, v, m, np.float32, 0)
    be.upload_x(x)
    got = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("LCX_PV_MFMA", flag)
        be.set_ws(w)
        be.moments_a(0); be.moments_b(0, 0.36, 1); be.moments_c(0)
        st = be.read_state(0)
        assert st[2] == 0
        arrays = {k: be.get_moment(0, k) for k in ("rho", "rhoinvrho", "Qij", "Si", "Qi-Si^2")}
        be.update_a()
        out = be.iterate(0.36, 1e-5, st[0], False)
        arrays["grad"] = be.get_moment(0, "grad")
        arrays["update"] = be.get_moment(0, "update")
        got[flag] = (st[0], out[1], out[2], int(out[3]), arrays)
    be.close()
    (tc0, tcn0, tan0, tr0, a0), (tc1, tcn1, tan1, tr1, a1) = got["0"], got["1"]
    assert abs(tc1 - tc0) < 2e-6 * max(1.0, abs(tc0)) and abs(tcn1 - tcn0) < 2e-5 * max(1.0, abs(tcn0)) and tr0 == tr1
    assert abs(tan1 - tan0) < 2e-5 * abs(tan0)
    for k in a0:
        assert relerr(a1[k], a0[k]) < 2e-5, (k, relerr(a1[k], a0[k]))
    for ls in ("exact", "linear"):
        hist = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("LCX_PV_MFMA", flag)
            mdl = Corex(n_hidden=m, seed=0, dtype=np.float32, device=0, max_iter=5, tol=0.0, line_search=ls).fit(x)
            hist[flag] = np.asarray(mdl.history["TC"], np.float64)
            mdl._backend.close()
        assert len(hist["0"]) == len(hist["1"]) == 35
        assert np.max(np.abs(hist["1"] - hist["0"]) / np.maximum(1.0, np.abs(hist["0"]))) < 5e-4, ls
