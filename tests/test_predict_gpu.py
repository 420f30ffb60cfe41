"""`predict` / `invert` on the device (lcx_predict / lcx_invert; reference linearcorex.py:431-441, g_inv :490-494).

Against tests/golden/g9_predict.npz - the reference's own outputs on big5 (tests/golden/make_golden_predict.py) - and the
oracle.  Kernel level (the fixture's X_i Z_j and theta handed to the library, as a restored model does): float64 1e-12 of
the output scale, float32 2e-5; end to end (our own fit of the same data, then predict of the reference's y): float64 1e-6,
float32 2e-3 (the float32 fits themselves differ by that much, tests/test_parity_gpu.py)."""
import pickle

import numpy as np
import pytest

from oracle import corex_oracle as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DT = {"f32": np.float32, "f64": np.float64}
KERNEL_TOL = {"f32": 2e-5, "f64": 1e-12}
E2E_TOL = {"f32": 2e-3, "f64": 1e-6}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, float(np.max(np.abs(b)))))


def _restored(g, p, gz, branch, dtype):
    """A model as `pickle.load` returns it: host arrays only, no device handle."""
    from linearcorex_amd import Corex
    mdl = Corex(n_hidden=5, gaussianize=gz, discourage_overlap=(branch == "ns"), dtype=dtype, device=0)
    mdl.ws = np.asarray(g[p + "ws"], dtype)
    mdl.nv, mdl.n_samples = 50, 2000
    mdl.theta = (g[p + "theta_mean"].astype(dtype), g[p + "theta_std"].astype(dtype))
    mdl.moments = {"X_i Z_j": np.asarray(g[p + "xz"], dtype)}
    return pickle.loads(pickle.dumps(mdl))


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("branch", ["ns", "syn"])
@pytest.mark.parametrize("gz", ["standard", "outliers"])
def test_predict_and_invert_of_a_restored_model(gz, branch, tag):
    g = load_golden("g9_predict")
    p = "%s_%s_%s_" % (gz, branch, tag)
    dt = DT[tag]
    mdl = _restored(g, p, gz, branch, dt)
    pred = mdl.predict(g[p + "y"])
    assert pred.shape == (97, 50) and pred.dtype == dt
    assert _rel(pred, g[p + "predict"]) < KERNEL_TOL[tag]
    assert _rel(mdl.predict(g[p + "y"][3]), g[p + "predict"][3]) < KERNEL_TOL[tag]           # one sample, 1-D like the reference
    # invert: finite where the reference is finite (float32 hits the poles of g_inv at |z| >= 5), equal there
    inv, ref = mdl.invert(g["z"]), g[p + "invert"]
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(inv), ok)
    assert _rel(inv[ok], ref[ok]) < KERNEL_TOL[tag]
    mdl._backend.close()


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("gz", ["standard", "outliers"])
def test_predict_after_fit_uses_the_resident_moments(gz, tag, g1):
    """fit on the device, then predict from the moments that are still resident (no X_i Z_j upload): against the reference's
    predict of the same latent factors, and against the oracle applied to this model's own moments."""
    from linearcorex_amd import Corex
    g = load_golden("g9_predict")
    p = "%s_ns_%s_" % (gz, tag)
    dt = DT[tag]
    mdl = Corex(n_hidden=5, seed=0, gaussianize=gz, dtype=dt, device=0).fit(g1["x_raw"])
    y = g[p + "y"]
    pred = mdl.predict(y)
    assert _rel(pred, g[p + "predict"]) < E2E_TOL[tag]
    own = O.predict(mdl.moments["X_i Z_j"].astype(np.float64), y, (mdl.theta[0].astype(np.float64), mdl.theta[1].astype(np.float64)), gz)
    assert _rel(pred, own) < KERNEL_TOL[tag] * 10
    # many rows: several 64-row tiles and a ragged last one
    yy = mdl.transform(g1["x_raw"])
    big = mdl.predict(yy)
    assert big.shape == (2000, 50)
    assert _rel(big[:97], mdl.predict(yy[:97])) == 0.0
    mdl._backend.close()


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_predict_synergistic_after_fit(tag):
    from linearcorex_amd import Corex
    x, _ = O.gen_planted(400, 300, 5, seed=4)
    dt = DT[tag]
    mdl = Corex(n_hidden=5, seed=0, discourage_overlap=False, dtype=dt, device=0, max_iter=50).fit(x)
    y = mdl.transform(x)[:130]
    pred = mdl.predict(y)
    own = O.predict(mdl.moments["X_i Z_j"].astype(np.float64), y.astype(np.float64),
                    (mdl.theta[0].astype(np.float64), mdl.theta[1].astype(np.float64)), "standard")
    assert _rel(pred, own) < KERNEL_TOL[tag] * 10
    # pickle -> transform -> get_covariance / predict on a synergistic model (the handle of a restored model holds no data)
    cov = mdl.get_covariance()
    back = pickle.loads(pickle.dumps(mdl))
    assert _rel(back.transform(x)[:130], y) < KERNEL_TOL[tag] * 10
    assert _rel(back.get_covariance(), cov) < KERNEL_TOL[tag] * 10
    assert _rel(back.predict(y), pred) < KERNEL_TOL[tag] * 10
    mdl._backend.close()
    back._backend.close()


def test_predict_wide_output_blocks():
    """more variables than one 64 MB staging block holds rows for: several row blocks, double-buffered copies"""
    from linearcorex_amd import Corex
    rng = np.random.RandomState(2)
    n, v, m = 3000, 9000, 12
    mdl = Corex(n_hidden=m, dtype=np.float32, device=0)
    mdl.ws = rng.randn(m, v).astype(np.float32)
    mdl.nv, mdl.n_samples = v, n
    mdl.theta = (rng.randn(v).astype(np.float32), (0.5 + rng.rand(v)).astype(np.float32))
    mdl.moments = {"X_i Z_j": (rng.randn(v, m) / 3).astype(np.float32)}
    y = rng.randn(n, m).astype(np.float32)
    pred = mdl.predict(y)
    ref = O.predict(mdl.moments["X_i Z_j"].astype(np.float64), y.astype(np.float64), mdl.theta, "standard")
    assert pred.shape == (n, v) and _rel(pred, ref) < 2e-5
    mdl._backend.close()


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("branch", ["ns", "syn"])
@pytest.mark.parametrize("gz", ["standard", "outliers"])
def test_transform_details_evaluates_the_new_batch(g1, gz, branch, tag):
    """`transform(x_new, details=True)` (reference :386-395) on the device: the batch goes through lcx_upload_preprocess with the
    fitted theta on a handle of its own, the levels of a full evaluation run with the fitted W and the FIT's sample count as the
    divisor (lcx_set_sample_divisor) - against the reference's own output, g10_transform_details.npz.  float64: the model is fitted
    here, 1e-6.  float32: the model is restored from the reference's fitted state (what an unpickled model holds), so that the
    evaluated path itself is compared at the float32 step-level bar instead of inheriting the drift of a float32 fit."""
    from linearcorex_amd import Corex
    from tests.test_host_logic_cpu import check_transform_details
    dt = np.float64 if tag == "f64" else np.float32
    mdl = check_transform_details(lambda gz_, ov: Corex(n_hidden=5, seed=0, dtype=dt, device=0, gaussianize=gz_, discourage_overlap=ov),
                                  g1, gz, branch, tag, 1e-6 if tag == "f64" else 1e-3, from_fixture=(tag == "f32"))
    mdl._backend.close()


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_pick_n_hidden_on_device(g1, tag):
    """The reference's exported helper `pick_n_hidden` (:458-480), a caller of the fit path, on the device: the scan over 1, 2, ...
    factors stops where the reference's stops, with its scores (float64 1e-6; float32 at the end-to-end bar of a float32 fit)."""
    from tests.test_host_logic_cpu import check_pick_n_hidden
    check_pick_n_hidden(g1, tag, 1e-6 if tag == "f64" else 2e-3, dtype=np.float64 if tag == "f64" else np.float32, device=0)
