"""Degenerate inputs the reference accepts without complaint: more factors than variables, a handful of samples, a constant
column (its standard deviation is clipped at 1e-10, reference :413), a single variable, more factors than samples.
float64 device fits must follow the oracle step for step; NaN cells without `missing_values` poison TC on both sides."""
import numpy as np
import pytest

from oracle import corex_oracle as O
from tests.test_parity_gpu import relerr

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.RandomState(3)
    const = rng.randn(100, 12)
    const[:, 4] = 2.5
    return {
        "more_factors_than_variables": (rng.randn(200, 5), 8),
        "three_samples": (rng.randn(3, 10), 2),
        "constant_column": (const, 3),
        "one_variable": (rng.randn(50, 1), 1),
        "more_factors_than_samples": (rng.randn(17, 200), 33),
    }


@pytest.mark.parametrize("name", sorted(_cases()))
def test_degenerate_inputs_follow_the_oracle(name):
    from linearcorex_amd import Corex
    x, m = _cases()[name]
    ref = O.fit_ns(x, m, seed=0, dtype=np.float64, max_iter=8, keep_x=True)
    out = Corex(n_hidden=m, seed=0, dtype=np.float64, device=0, max_iter=8).fit(x)
    h, hr = np.asarray(out.history["TC"], np.float64), np.asarray(ref.history_tc, np.float64)
    assert len(h) == len(hr), (len(h), len(hr))
    # TC of a structureless one-variable problem is rounding noise around 0: absolute bar there
    assert np.max(np.abs(h - hr)) < 1e-9 * max(1.0, float(np.max(np.abs(hr)))) + 1e-12
    assert out.ws.shape == ref.ws.shape and np.all(np.isfinite(out.ws))
    if name != "one_variable":
        assert np.max(np.abs(out.ws - ref.ws)) < 1e-7 * max(1.0, float(np.max(np.abs(ref.ws))))
        assert np.array_equal(out.clusters(), ref.clusters())
    assert relerr(out.get_covariance(), ref.get_covariance()) < 1e-6
    assert relerr(out.transform(x), ref.transform(ref.x_tilde)) < 1e-6 or np.max(np.abs(out.transform(x))) < 1e-9


def test_nan_cells_without_missing_values_poison_tc_like_the_reference(capsys):
    """The reference does not look for NaN unless `missing_values` is set: TC becomes NaN, `fit` prints its error line per
    iteration (:146) and carries on to max_iter.  Same here."""
    from linearcorex_amd import Corex
    x = np.random.RandomState(4).randn(60, 9)
    x[5, 2] = np.nan
    out = Corex(n_hidden=2, seed=0, dtype=np.float64, device=0, max_iter=3).fit(x)
    assert len(out.history["TC"]) == 21 and not np.isfinite(out.history["TC"][-1])
    assert "TC is no longer finite" in capsys.readouterr().out


def test_out_of_memory_in_create_is_an_error_and_leaks_nothing():
    """A shard that cannot be allocated (X and its transposed copy alone would be 2 x 16 TB) fails with LcxError, releases what
    it had allocated (the same request can be repeated without the device filling up) and leaves the library usable."""
    import ctypes as C
    from linearcorex_amd import Corex, _abi
    from linearcorex_amd.backend import HipBackend
    lib = _abi.load()
    free0, total = C.c_size_t(), C.c_size_t()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo(C.byref(free0), C.byref(total))
    for _ in range(3):
        with pytest.raises(_abi.LcxError, match="hipMalloc|out of memory|memory"):
            HipBackend(2_000_000, 2_000_000, 64, np.float32, 0)
    free1 = C.c_size_t()
    hip.hipMemGetInfo(C.byref(free1), C.byref(total))
    assert free0.value - free1.value < (256 << 20), (free0.value, free1.value)
    x, _ = O.gen_planted(200, 60, 3, seed=5)
    out = Corex(n_hidden=3, seed=0, dtype=np.float64, device=0, max_iter=5).fit(x)
    ref = O.fit_ns(x, 3, seed=0, dtype=np.float64, max_iter=5)
    assert np.max(np.abs(np.asarray(out.history["TC"], np.float64) - np.asarray(ref.history_tc))) < 1e-9
    assert lib.lcx_abi_version() == 1


def test_abi_misuse_returns_status_codes():
    """Plain-C callers get status codes, never a crash: null handles and pointers, out-of-range arguments, calls out of
    sequence (the error conventions table of INTEGRATION.md)."""
    import ctypes as C
    from linearcorex_amd import _abi
    from linearcorex_amd.backend import HipBackend
    lib = _abi.load()
    ARG, STATE = 1, 4
    assert lib.lcx_moments_a(None, 0) == ARG and b"null handle" in lib.lcx_last_error()
    assert lib.lcx_destroy(None) == 0
    h = C.c_void_p()
    assert lib.lcx_create(C.byref(h), 0, 10, 2, 0, 0) == ARG                 # no samples
    assert lib.lcx_create(C.byref(h), 10, 10, 2, 7, 0) == ARG                # unknown dtype
    assert lib.lcx_create(C.byref(h), 10, 10, 2, 0, 99) == ARG               # no such device
    assert lib.lcx_create(C.byref(h), 10, 10, 1025, 0, 0) == ARG             # more than 1024 factors
    be = HipBackend(64, 40, 3, np.float64, 0)
    assert lib.lcx_moments_a(be.h, 2) == ARG                                  # which must be 0 or 1
    assert lib.lcx_make_trial(be.h, C.c_double(1.0)) == STATE                 # no direction yet
    assert lib.lcx_set_ws(be.h, None) == ARG
    assert lib.lcx_upload_x(be.h, None, 40) == ARG
    assert lib.lcx_read_state(be.h, 0, None) == ARG
    out = (C.c_double * 8)()
    assert lib.lcx_covariance_rows(be.h, C.c_double(0.0), None, 0, 1, None) == ARG
    assert lib.lcx_covariance_rows(be.h, C.c_double(0.0), C.cast(out, C.c_void_p), 39, 5, C.cast(out, C.c_void_p)) == ARG   # rows past the end
    assert lib.lcx_timing_read(be.h, 7, None, None) == ARG and lib.lcx_timing_read(be.h, -1, None, None) == ARG      # sites 0 .. 6
    assert lib.lcx_timing_read(be.h, 6, None, None) == 0
    assert lib.lcx_syn_update_a(be.h) == STATE                                # before any synergistic moments
    buf = C.create_string_buffer(8)
    assert lib.lcx_kernel_name(be.h, 0, buf, 8) == ARG                        # buffer too small
    # the handle is still good after all of that
    x = np.random.RandomState(0).randn(64, 40)
    be.upload_x(x - x.mean(0))
    be.set_ws(np.random.RandomState(1).randn(3, 40) * 0.01)
    be.moments_a(0); be.moments_b(0, 0.0, 0); be.moments_c(0)
    assert np.isfinite(be.read_state(0)[0])
    be.close()


def test_repeated_fits_release_their_device_memory():
    """Fit / get_covariance / transform / close in a loop (both branches, merged-pass buffers, covariance staging, synergistic
    buffers): free device memory after the loop is what it was before it, and refitting one object with another shape works."""
    import ctypes as C
    from linearcorex_amd import Corex
    hip = C.CDLL("libamdhip64.so")
    free0, free1, total = C.c_size_t(), C.c_size_t(), C.c_size_t()
    rng = np.random.RandomState(0)
    warm = Corex(n_hidden=3, seed=0, dtype=np.float32, device=0, max_iter=2).fit(rng.randn(100, 50))
    warm.get_covariance()
    warm._backend.close()
    import gc
    gc.collect()                                         # handles of earlier tests that were never closed go now, not mid-loop
    hip.hipMemGetInfo(C.byref(free0), C.byref(total))
    for k in range(12):
        syn = k % 3 == 2
        x = rng.randn(300 + 37 * k, 200 + 61 * k)
        mdl = Corex(n_hidden=4 + k, seed=0, dtype=np.float32 if k % 2 else np.float64, device=0, max_iter=3,
                    discourage_overlap=not syn).fit(x)
        assert mdl.get_covariance().shape == (x.shape[1], x.shape[1])
        assert mdl.transform(x[:17]).shape == (17, 4 + k)
        if k % 4 == 0:                                   # the same object again, another shape
            x2 = rng.randn(150, 90)
            mdl.ws = np.zeros((0, 0))
            mdl.fit(x2)
            assert mdl.ws.shape == (4 + k, 90)
        mdl._backend.close()
    hip.hipMemGetInfo(C.byref(free1), C.byref(total))
    assert int(free0.value) - int(free1.value) < (64 << 20), (free0.value, free1.value)      # a leak shows as LESS free memory


def test_empirical_gaussianize_end_to_end(capsys):
    """gaussianize='empirical' (reference :424-426: rank transform per column, here on the device) must give the fit of the
    rank-transformed data, and `transform` of a new batch ranks that batch (as the reference's preprocess does); 'none' (and
    any unknown name, :404-405) passes the data through."""
    from linearcorex_amd import Corex
    x = np.random.RandomState(9).lognormal(size=(300, 40))
    x[:, 3] = np.round(x[:, 3])                                                          # ties
    out = Corex(n_hidden=3, seed=0, dtype=np.float64, device=0, max_iter=6, gaussianize="empirical").fit(x)
    assert "empirical gauss transform not implemented" in capsys.readouterr().out      # the reference's own warning (:425)
    xr = O.preprocess(x.copy(), None, "empirical")[0]
    ref = Corex(n_hidden=3, seed=0, dtype=np.float64, device=0, max_iter=6, gaussianize="none").fit(xr)
    h, hr = np.asarray(out.history["TC"], np.float64), np.asarray(ref.history["TC"], np.float64)
    assert len(h) == len(hr) and np.max(np.abs(h - hr)) < 1e-9
    orc = O.fit_ns(xr, 3, seed=0, dtype=np.float64, max_iter=6, gaussianize="none")
    assert np.max(np.abs(h - np.asarray(orc.history_tc))) < 1e-9
    assert np.max(np.abs(out.transform(x) - xr.dot(orc.ws.T))) < 1e-8
    x2 = np.random.RandomState(10).lognormal(size=(77, 40))
    assert np.max(np.abs(out.transform(x2) - O.preprocess(x2.copy(), None, "empirical")[0].dot(orc.ws.T))) < 1e-8
    assert out.invert(xr[:5]) is not None and np.array_equal(out.invert(xr[:5]), xr[:5])    # no inverse: passes through (:437-438)
    out._backend.close()
    ref._backend.close()
