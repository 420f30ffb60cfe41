"""The arithmetic claim behind `f32_gemm="split"` (linearcorex_amd/csrc/gemm_split_kernels.hpp), checked in NumPy with
the device code's own operations: a float32 number IS the sum of three bf16 numbers (round-to-nearest split: hi = bf16(x), mid =
bf16(x - hi), lo = x - hi - mid - v_cvt_pk_bf16_f32 rounds to nearest even), a bf16 x bf16 product is exact in float32, and the 3 partial
products the kernel drops are zero-mean, 2^-27 |a b| in the rms and at most 2^-24.  (Round 4 shipped a truncation split - mask the upper
16 bits -, whose dropped terms all carry the sign of the product: a systematic shrink of up to 2^-21; kept below as the comparison.)
No GPU, no library: this pins the algorithm, the GPU tests pin the kernel."""
import numpy as np
import pytest

MASK = np.uint32(0xFFFF0000)


def rne_bf16(x):
    """v_cvt_pk_bf16_f32 element-wise: float32 -> the nearest bf16 number (ties to even), as a float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    """split8 of the device code, element-wise: three float32 arrays whose low 16 bits are zero (= bf16 numbers)."""
    x = np.asarray(x, np.float32)
    hi = rne_bf16(x)
    r1 = x - hi                                   # float32 subtraction, as on the device (exact)
    mid = rne_bf16(r1)
    lo = r1 - mid                                 # exact, and at most 8 significant bits are left
    return hi, mid, lo


def split3_truncating(x):
    """round 4's split (no longer in the library): hi = the upper 16 bits of the word, mid = the upper 16 bits of x - hi"""
    x = np.asarray(x, np.float32)
    hi = (x.view(np.uint32) & MASK).view(np.float32)
    r1 = x - hi
    mid = (r1.view(np.uint32) & MASK).view(np.float32)
    return hi, mid, r1 - mid


def samples(n, seed):
    rng = np.random.RandomState(seed)
    x = rng.randn(n).astype(np.float32)
    x[: n // 8] *= np.float32(1e-3)
    x[n // 8: n // 4] *= np.float32(1e4)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.4e38, -3.4e38, 1.17549435e-38, 2.0 ** -126 * 1.5,
                     1e-30, 224.0, 65504.0, 0.1, -0.3333333], np.float32)
    return np.concatenate([x, edge])


def test_a_float32_is_exactly_three_bf16_numbers():
    x = samples(200000, 0)
    # the documented caveat of rounding to nearest: within half a bf16 ulp of FLT_MAX hi rounds to infinity (the library's operands
    # are standardised data and weights of order 1)
    big = np.abs(x) >= np.float32(3.38e38)
    with np.errstate(invalid="ignore"):          # (x - inf) + ... is the NaN the caveat is about
        assert np.all(np.isinf(split3(x[big])[0])) and big.sum() == 2
    x = x[~big]
    hi, mid, lo = split3(x)
    for part in (hi, mid, lo):
        assert not np.any(part.view(np.uint32) & np.uint32(0x0000FFFF))          # bf16-representable: nothing below bit 16
    # exact: the sum in float64 is the number itself (every subtraction above was exact)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # the residuals are at most half a bf16 ulp of what they are taken from, and signed (zero-mean) ...
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(mid) <= ax * 2.0 ** -8) and np.all(np.abs(lo) <= ax * 2.0 ** -16)
    nz = (mid != 0) & (np.abs(x) > 1e-30)
    assert 0.4 < np.mean(np.signbit(mid[nz]) == np.signbit(x[nz])) < 0.6
    # ... where the truncation split's all carry the sign of x (the bias the library no longer has)
    th, tm, tl = split3_truncating(x)
    assert np.array_equal(th.astype(np.float64) + tm + tl, x.astype(np.float64))
    assert np.all(np.signbit(tm[tm != 0]) == np.signbit(x[tm != 0]))


def test_the_kept_products_are_exact_in_float32_and_the_dropped_ones_below_half_a_rounding():
    a, b = samples(100000, 1), samples(100000, 2)
    keep = ~(np.isinf(a.astype(np.float64) * b.astype(np.float64)) | (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 3e38)
             | (np.abs(a) > 1e38) | (np.abs(b) > 1e38))
    a, b = a[keep], b[keep]
    tiny = np.abs(a.astype(np.float64) * b.astype(np.float64)) < 1e-30            # products near the float32 underflow range: not the claim
    a, b = a[~tiny], b[~tiny]
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    kept = [(am, bh), (ah, bm), (ah, bh), (al, bh), (ah, bl), (am, bm)]          # SPLIT_PA / SPLIT_PB with NP = 6
    total = np.zeros(a.shape, np.float64)
    for p, q in kept:
        prod32 = p * q                                                            # float32 product of two bf16 numbers ...
        assert np.array_equal(prod32.astype(np.float64), p.astype(np.float64) * q.astype(np.float64))   # ... is exact (16 significand bits)
        total += prod32.astype(np.float64)
    exact = a.astype(np.float64) * b.astype(np.float64)
    dropped = exact - total                                                       # = am bl + al bm + al bl
    rel = dropped / exact
    assert np.all(np.abs(rel) < 2.0 ** -24)                                       # |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|: 2 x 2^-25 + 2^-34
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -27 and abs(np.mean(rel)) < 2.0 ** -30      # below half a float32 rounding, zero-mean
    # round 4's truncation split on the same numbers: 8 x larger in the worst case and of ONE sign (a shrink of every product)
    th, tm, tl = split3_truncating(a)
    uh, um, ul = split3_truncating(b)
    t_total = sum(p.astype(np.float64) * q.astype(np.float64) for p, q in ((tm, uh), (th, um), (th, uh), (tl, uh), (th, ul), (tm, um)))
    t_rel = (exact - t_total) / exact
    assert np.all(t_rel >= 0) and np.max(t_rel) > 2.0 ** -22.5 and np.mean(t_rel) > 2.0 ** -25.5


@pytest.mark.parametrize("k", [4096, 100032])
def test_a_contraction_with_six_products_is_float32_grade(k):
    """Dot products of length k accumulated in float32 from the 6 kept partial products, against float64: the error is that of a
    float32 dot product (the dropped terms do not show), far from that of 3 products."""
    rng = np.random.RandomState(3)
    a = rng.randn(64, k).astype(np.float32)
    b = rng.randn(k).astype(np.float32)
    ref = a.astype(np.float64).dot(b.astype(np.float64))
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)

    def dot32(pairs):
        acc = np.zeros(64, np.float32)
        for c0 in range(0, k, 32):                                                # one MFMA step = 32 contraction elements
            for p, q in pairs:
                acc = (acc + (p[:, c0:c0 + 32].astype(np.float64) * q[c0:c0 + 32].astype(np.float64)).sum(1).astype(np.float32)).astype(np.float32)
        return acc
    six = dot32([(am, bh), (ah, bm), (ah, bh), (al, bh), (ah, bl), (am, bm)])
    three = dot32([(am, bh), (ah, bm), (ah, bh)])
    plain = np.zeros(64, np.float32)
    for c0 in range(0, k, 4):                                                     # the float32 MFMA: 4 elements per step
        plain = (plain + (a[:, c0:c0 + 4].astype(np.float64) * b[c0:c0 + 4].astype(np.float64)).sum(1).astype(np.float32)).astype(np.float32)
    scale = np.sqrt(np.mean(ref ** 2))
    e6, e3, e1 = (np.sqrt(np.mean((v.astype(np.float64) - ref) ** 2)) / scale for v in (six, three, plain))
    # (with the round-to-nearest split even 3 products come within ~2-5 x of float32 grade: residuals are half the truncation split's)
    assert e6 < 1.5 * e1 + 1e-7 and e3 > 1.5 * e6, (e6, e3, e1)
