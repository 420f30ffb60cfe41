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
