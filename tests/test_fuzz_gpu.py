"""Randomised end-to-end parity (float64, seeded): odd sizes around every padding granule, several n_hidden,
both branches and both line-search modes, against the oracle.  Same iteration counts, histories within 1e-6."""
import numpy as np
import pytest

from oracle import corex_oracle as O
from tests.test_parity_gpu import relerr

pytestmark = pytest.mark.gpu

rng = np.random.RandomState(2024)
CASES = []
for _ in range(14):
    n = int(rng.choice([33, 64, 65, 127, 200, 513, 1000]))
    v = int(rng.choice([2, 5, 15, 16, 17, 63, 64, 65, 129, 300, 1023]))
    m = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 64, 100]))
    if m > v:
        m = max(1, v // 2)
    CASES.append((n, v, m, int(rng.randint(1, 1000))))


@pytest.mark.parametrize("n,v,m,seed", CASES)
def test_random_shapes_ns(n, v, m, seed):
    from linearcorex_amd import Corex
    x, _ = O.gen_planted(n, v, max(1, min(m, 4)), seed=seed)
    ref = O.fit_ns(x, m, seed=0, dtype=np.float64, max_iter=12)
    for mode in ("exact", "linear"):
        out = Corex(n_hidden=m, seed=0, dtype=np.float64, device=0, max_iter=12, line_search=mode).fit(x)
        h, hr = np.asarray(out.history["TC"], np.float64), np.asarray(ref.history_tc)
        assert len(h) == len(hr), (mode, len(h), len(hr))
        assert relerr(h, hr) < 1e-6, mode
        assert relerr(out.ws, ref.ws) < 1e-5, mode
        assert relerr(out.get_covariance(), ref.get_covariance()) < 1e-6, mode


@pytest.mark.parametrize("n,v,m,seed", CASES[:6])
def test_random_shapes_syn(n, v, m, seed):
    from linearcorex_amd import Corex
    x, _ = O.gen_planted(n, v, max(1, min(m, 4)), seed=seed)
    ref = O.fit_syn(x, m, seed=0, dtype=np.float64, max_iter=40)
    out = Corex(n_hidden=m, seed=0, dtype=np.float64, device=0, max_iter=40, discourage_overlap=False).fit(x)
    h, hr = np.asarray(out.history["TC"], np.float64), np.asarray(ref.history_tc)
    assert len(h) == len(hr)
    assert relerr(h, hr) < 1e-6
    assert relerr(out.get_covariance(), ref.get_covariance()) < 1e-6
