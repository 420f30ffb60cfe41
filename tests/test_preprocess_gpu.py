"""Parity of the on-device preprocess (lcx_upload_preprocess / lcx_project_raw; reference
linearcorex.py:397-429, mean_impute :497-510, g :483-487) against the oracle's NumPy restatement.

float64: 1e-12 of the data scale; float32: 2e-6 (the device accumulates column sums in double and rounds
theta to the working dtype, NumPy sums pairwise in float32).  n_obs (integer output): bit-exact."""
import numpy as np
import pytest

from oracle import corex_oracle as O

pytestmark = pytest.mark.gpu
TOL = {np.float32: 2e-6, np.float64: 1e-12}
THETA_TOL = {np.float32: 1e-5, np.float64: 1e-12}     # NumPy's own float32 column means carry ~3e-6


def _raw(n, v, seed):
    rng = np.random.RandomState(seed)
    x = rng.randn(n, v) * (0.5 + 3 * rng.rand(v)) + 10 * rng.randn(v)
    heavy = np.arange(v) % 7 == 0
    x[:, heavy] = np.sign(x[:, heavy]) * np.abs(x[:, heavy]) ** 1.7
    return x


def _backend(n, v, dtype, m=3):
    from linearcorex_amd.backend import HipBackend
    return HipBackend(n, v, m, dtype, 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["standard", "outliers", "none"])
@pytest.mark.parametrize("shape", [(300, 70), (1000, 257), (64, 1000), (4097, 130)])
def test_preprocess_fit(dtype, kind, shape):
    n, v = shape
    x = _raw(n, v, 11).astype(dtype)
    ref, theta_ref, nobs_ref = O.preprocess(x.copy(), gaussianize=kind)
    be = _backend(n, v, dtype)
    theta, n_obs, max_abs = be.upload_preprocess(x, kind, None, None)
    got = be.download_x()
    scale = max(1.0, float(np.max(np.abs(ref))))
    assert np.max(np.abs(got.astype(np.float64) - np.asarray(ref, np.float64))) < THETA_TOL[dtype] * scale * 10
    assert n_obs == n == nobs_ref
    if kind != "none":
        assert np.max(np.abs(theta[0] - theta_ref[0]) / np.maximum(1, np.abs(theta_ref[0]))) < THETA_TOL[dtype]
        assert np.max(np.abs(theta[1] - theta_ref[1]) / theta_ref[1]) < THETA_TOL[dtype] * 10
        assert abs(max_abs - float(np.max(np.abs(ref)))) < 1e-4 * scale
    be.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["standard", "outliers"])
@pytest.mark.parametrize("sentinel", [-1e6, float("nan")])
def test_preprocess_missing_values(dtype, kind, sentinel, g6):
    x = np.array(g6["x_raw"], dtype=dtype)                  # adni_blood: -1e6 marks missing cells
    if np.isnan(sentinel):
        x[x == -1e6] = np.nan
    n, v = x.shape
    ref, theta_ref, nobs_ref = O.preprocess(x.copy(), gaussianize=kind, missing_values=sentinel)
    be = _backend(n, v, dtype)
    theta, n_obs, _ = be.upload_preprocess(x, kind, sentinel, None)
    assert np.array_equal(n_obs, nobs_ref)                  # integer output: bit-exact
    got = be.download_x()
    # float32: the reference's column mean (float32 pairwise sum over the imputed column) differs from the
    # mean of the observed cells it imputed with by ~1e-4 of a std on this data (imputed cells come out as
    # 2e-4, not 0); the device imputes and centres with one double-accumulated mean.  float64: 1e-10.
    tol = 1e-3 if dtype == np.float32 else 1e-10
    assert np.max(np.abs(got.astype(np.float64) - np.asarray(ref, np.float64))) < tol
    assert np.max(np.abs(theta[1] - theta_ref[1]) / theta_ref[1]) < (1e-4 if dtype == np.float32 else 1e-12)
    # transform-time preprocess of other data with the fitted theta (fit == 0)
    x2 = x[::2].copy()
    ref2 = O.preprocess(x2.copy(), theta=theta_ref, gaussianize=kind, missing_values=sentinel)[0]
    be2 = _backend(x2.shape[0], v, dtype)
    be2.upload_preprocess(x2, kind, sentinel, theta_ref)
    assert np.max(np.abs(be2.download_x().astype(np.float64) - np.asarray(ref2, np.float64))) < tol
    be.close(); be2.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["standard", "outliers", "none"])
def test_project_raw_matches_host_preprocess(dtype, kind):
    n, v, m = 500, 333, 6
    x = _raw(n, v, 5).astype(dtype)
    be = _backend(n, v, dtype, m)
    theta, _, _ = be.upload_preprocess(x, kind, None, None)
    w = (np.random.RandomState(1).randn(m, v) * 0.05).astype(dtype)
    be.set_ws(w)
    xn = _raw(9000, v, 6).astype(dtype)                      # more rows than one staged block
    y = be.project_raw(xn, kind, theta)
    xt = O.preprocess(xn.copy(), theta=theta, gaussianize=kind)[0]
    ref = np.asarray(xt, np.float64) @ w.astype(np.float64).T
    scale = np.abs(np.asarray(xt, np.float64)) @ np.abs(w.astype(np.float64)).T
    assert np.max(np.abs(y - ref) / scale) < (2e-5 if dtype == np.float32 else 1e-12)
    be.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("name", ["emp", "emp_missing"])
def test_preprocess_empirical_against_the_reference(dtype, name):
    """gaussianize='empirical' on the device (lcx_upload_preprocess kind 3: segmented sort of the transposed copy, average ranks
    for ties, AS 241 normal quantile) against the reference's own preprocess output (g9_predict.npz; heavy ties, continuous
    columns, missing cells imputed before ranking).  Ranks are integers / half-integers: an error there would show as >= 1e-3."""
    from tests.conftest import load_golden
    g = load_golden("g9_predict")
    tag = "f32" if dtype == np.float32 else "f64"
    x = np.array(g["emp_x_missing" if name == "emp_missing" else "emp_x"], dtype=dtype)
    ref = g["%s_%s" % (name, tag)]
    n, v = x.shape
    be = _backend(n, v, dtype)
    theta, n_obs, _ = be.upload_preprocess(x, "empirical", -1e6 if name == "emp_missing" else None, None)
    got = be.download_x()
    assert np.max(np.abs(got.astype(np.float64) - ref)) < (1e-12 if dtype == np.float64 else 5e-7)
    if name == "emp_missing":
        assert np.array_equal(n_obs, np.sum(x != -1e6, axis=0))
    # both layouts of the shard carry the transformed data: X.W^T through the transposed copy
    w = np.random.RandomState(0).randn(3, v).astype(dtype)
    be.set_ws(w)
    y = be.project_resident()
    assert np.max(np.abs(y - ref.astype(dtype).dot(w.T))) < (1e-10 if dtype == np.float64 else 2e-4)
    be.close()


@pytest.mark.parametrize("shape", [(64, 70), (1000, 257), (5000, 33), (70000, 5)])
def test_preprocess_empirical_shapes(shape):
    """ragged sizes, more rows than one sort chunk holds columns for, all-equal and two-valued columns"""
    n, v = shape
    rng = np.random.RandomState(n + v)
    x = rng.randn(n, v)
    x[:, 0] = 3.0                                  # one tie group of n
    x[:, 1] = (rng.rand(n) < 0.3) * 1.0            # two values
    x[:, 2] = np.round(x[:, 2], 1)
    ref = O.preprocess(x.copy(), None, "empirical")[0]
    be = _backend(n, v, np.float64)
    be.upload_preprocess(x, "empirical", None, None)
    got = be.download_x()
    assert np.max(np.abs(got - ref)) < 1e-12
    be.close()


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("gz,missing,branch", [("standard", None, "ns"), ("outliers", None, "ns"), ("none", None, "ns"),
                                               ("empirical", None, "ns"), ("standard", -999.0, "ns"), ("standard", None, "syn")])
def test_fit_transform_reads_the_resident_data(gz, missing, branch, tag, monkeypatch):
    """`fit_transform(x)` (reference :103-105: fit, then transform of the same x) takes the latent factors from the shard
    that is still resident - one `lcx_moments_a` instead of a second upload and preprocess of x - and must give what
    `transform(x)` gives; against the oracle's x~ . ws^T as well."""
    import numpy as np
    from linearcorex_amd import Corex
    from linearcorex_amd.backend import HipBackend
    from oracle import corex_oracle as O
    dt = {"f32": np.float32, "f64": np.float64}[tag]
    x = np.random.RandomState(21).randn(333, 70)
    x[:, :20] += 1.5 * x[:, [0]]
    x = x * np.linspace(0.5, 3.0, 70) + 2.0
    if gz == "none":
        x = (x - x.mean(0)) / x.std(0)
    if missing is not None:
        x[np.random.RandomState(22).rand(*x.shape) < 0.04] = missing
    uploads = []
    real = HipBackend.project_raw
    monkeypatch.setattr(HipBackend, "project_raw", lambda self, *a, **k: (uploads.append(1), real(self, *a, **k))[1])
    mdl = Corex(n_hidden=4, seed=0, dtype=dt, device=0, max_iter=6, gaussianize=gz, missing_values=missing,
                discourage_overlap=(branch == "ns"))
    y = mdl.fit_transform(x)
    assert not uploads and y.shape == (333, 4) and y.dtype == dt
    y2 = mdl.transform(x)
    tol = 1e-12 if tag == "f64" else 2e-5
    scale = max(1.0, float(np.max(np.abs(y2))))
    assert np.max(np.abs(y.astype(np.float64) - y2)) < tol * scale
    xt = O.preprocess(x.astype(dt).astype(np.float64), None, gz, missing)[0]
    assert np.max(np.abs(y - xt.dot(np.asarray(mdl.ws, np.float64).T))) < (1e-10 if tag == "f64" else 1e-3) * scale
    mdl._backend.close()
