"""Parity of the two X-streaming MFMA kernels in isolation (through the C ABI test entry points)
against NumPy float64, for every padded n_hidden, both dtypes, ragged sizes and forced splits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {np.float32: 2e-5, np.float64: 1e-12}


def _asym(n, k, seed):
    rng = np.random.RandomState(seed)
    return rng.randn(n, k) + 0.01 * np.arange(k)[None, :] + 0.1 * np.arange(n)[:, None] / max(n, 1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m_pad", [16, 32, 64, 128])
@pytest.mark.parametrize("shape,split,waves", [((70, 45), 1, 1), ((257, 300), 1, 4), ((1000, 1111), 3, 4),
                                               ((64, 2048), 2, 8), ((333, 64), 1, 2)])
def test_gemm_nt(dtype, m_pad, shape, split, waves):
    from tests.probe import gemm_nt_check
    n, k = shape
    a = _asym(n, k, 1).astype(dtype)
    b = _asym(k, m_pad, 2).astype(dtype)          # asymmetric: catches row/col swaps
    got = gemm_nt_check(a, b, m_pad, dtype, split=split, waves=waves)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert np.max(np.abs(got - ref) / scale) < TOL[dtype]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m_pad", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("shape,split,waves", [((70, 45), 1, 1), ((300, 257), 1, 4), ((1111, 1000), 3, 4),
                                               ((2048, 64), 2, 8), ((64, 333), 1, 2)])
def test_gemm_tn(dtype, m_pad, shape, split, waves):
    from tests.probe import gemm_tn_check
    k, v = shape
    a = _asym(k, v, 3).astype(dtype)
    b = _asym(k, m_pad, 4).astype(dtype)
    got = gemm_tn_check(a, b, m_pad, dtype, split=split, waves=waves)
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    assert np.max(np.abs(got - ref) / scale) < TOL[dtype]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gemm_tn_rowscale(dtype):
    from tests.probe import gemm_tn_check
    k, v, m_pad = 500, 200, 32
    a = _asym(k, v, 5).astype(dtype)
    b = _asym(k, m_pad, 6).astype(dtype)
    s = np.random.RandomState(7).rand(k).astype(dtype) + 0.5
    got = gemm_tn_check(a, b, m_pad, dtype, rowscale=s, split=2, waves=4)
    ref = (a.astype(np.float64) * s.astype(np.float64)[:, None]).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64) * 1.5
    assert np.max(np.abs(got - ref) / scale) < TOL[dtype] * 2


def test_gemm_linearity_at_config2_size():
    """Size-independent property at BASELINE.json config-2 shape (10k x 5k, m=32, f64):
    (X.(B1+B2)^T) == X.B1^T + X.B2^T and X^T.(Y1+Y2) == X^T.Y1 + X^T.Y2 up to rounding."""
    from tests.probe import gemm_nt_check, gemm_tn_check
    rng = np.random.RandomState(0)
    n, v, m_pad = 10000, 5000, 32
    x = rng.randn(n, v)
    b1, b2 = rng.randn(v, m_pad), rng.randn(v, m_pad)
    y1 = gemm_nt_check(x, b1, m_pad, np.float64, split=3)
    y2 = gemm_nt_check(x, b2, m_pad, np.float64, split=3)
    y12 = gemm_nt_check(x, b1 + b2, m_pad, np.float64, split=3)
    assert np.max(np.abs(y12 - (y1 + y2))) < 1e-10 * np.sqrt(v)
    # spot check a few rows against NumPy
    rows = [0, 1, 4999, 9999]
    assert np.allclose(y1[rows], x[rows] @ b1, rtol=0, atol=1e-10)
    d1 = gemm_tn_check(x, y1, m_pad, np.float64, split=5)
    d2 = gemm_tn_check(x, y2, m_pad, np.float64, split=5)
    d12 = gemm_tn_check(x, y12, m_pad, np.float64, split=5)
    assert np.max(np.abs(d12 - (d1 + d2)) / np.abs(d12).max()) < 1e-12
    cols = [0, 17, 4999]
    assert np.allclose(d1[cols], x[:, cols].T @ y1, rtol=1e-11, atol=1e-7)


# ---- gemm_ct: column-tiled waves, B through LDS, stream-K balanced blocks --------------------------
# waves=-1 selects it in the test entry point; split = number of blocks (0: what production would launch).
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m_pad", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("shape,blocks", [((70, 45), 0), ((300, 257), 3), ((1111, 1000), 0), ((2048, 64), 7),
                                          ((64, 333), 1), ((1000, 1984), 37), ((4096, 640), 512)])
def test_gemm_ct(dtype, m_pad, shape, blocks):
    """Ragged sizes (inactive trailing waves of a super tile), block counts that cut super tiles at
    arbitrary groups (several partial slots + zero-filled ones), one block, more blocks than units."""
    from tests.probe import gemm_tn_check
    k, v = shape
    a = _asym(k, v, 8).astype(dtype)
    b = _asym(k, m_pad, 9).astype(dtype)
    got = gemm_tn_check(a, b, m_pad, dtype, split=blocks, waves=-1)
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)
    assert np.max(np.abs(got - ref) / scale) < TOL[dtype]


def test_gemm_ct_is_deterministic_and_matches_tn():
    from tests.probe import gemm_tn_check
    rng = np.random.RandomState(3)
    k, v, m_pad = 5000, 3000, 64
    a, b = rng.randn(k, v).astype(np.float32), rng.randn(k, m_pad).astype(np.float32)
    c1 = gemm_tn_check(a, b, m_pad, np.float32, split=0, waves=-1)
    c2 = gemm_tn_check(a, b, m_pad, np.float32, split=0, waves=-1)
    assert np.array_equal(c1, c2)                       # fixed summation order, no atomics
    t = gemm_tn_check(a, b, m_pad, np.float32, split=3, waves=4)
    assert np.max(np.abs(c1 - t)) < 2e-5 * np.sqrt(k) * 4


# ---- gemm_tn4: the float64 small-shard contraction on v_mfma_f64_4x4x4 (waves=-2 in the test entry point) ---------
@pytest.mark.parametrize("m_pad", [16, 32])
@pytest.mark.parametrize("shape,split", [((70, 45), 1), ((300, 257), 1), ((1111, 1000), 3), ((2048, 64), 2), ((64, 333), 1),
                                         ((4096, 640), 7), ((10000, 1300), 6)])
def test_gemm_tn4(m_pad, shape, split):
    from tests.probe import gemm_tn_check
    k, v = shape
    a = _asym(k, v, 10).astype(np.float64)
    b = _asym(k, m_pad, 11).astype(np.float64)
    got = gemm_tn_check(a, b, m_pad, np.float64, split=split, waves=-2)
    ref = a.T @ b
    scale = np.abs(a).T @ np.abs(b)
    assert np.max(np.abs(got - ref) / scale) < TOL[np.float64]
    # same result, different instruction: the 16x16x4 kernel
    alt = gemm_tn_check(a, b, m_pad, np.float64, split=split, waves=4)
    assert np.max(np.abs(got - alt) / scale) < TOL[np.float64]
