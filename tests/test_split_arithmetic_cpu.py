"""The arithmetic claim behind `f32_gemm="split"` (linearcorex_amd/csrc/gemm_split_kernels.hpp), checked in NumPy with
the device code's own bit operations: a float32 number IS the sum of three bf16 numbers (truncation split: hi = the upper 16 bits of the
word, mid = the upper 16 bits of x - hi, lo = x - hi - mid), a bf16 x bf16 product is exact in float32, and the 3 partial products the
kernel drops are 2^-24 |a b| in the rms and at most 2^-21.  No GPU, no library: this pins the algorithm, the GPU tests pin the kernel."""
import numpy as np
import pytest

MASK = np.uint32(0xFFFF0000)


def split3(x):
    """split8 of the device code, element-wise: three float32 arrays whose low 16 bits are zero (= bf16 numbers)."""
    x = np.asarray(x, np.float32)
    hi = (x.view(np.uint32) & MASK).view(np.float32)
    r1 = x - hi                                   # float32 subtraction, as on the device
    mid = (r1.view(np.uint32) & MASK).view(np.float32)
    lo = r1 - mid
    return hi, mid, lo


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
    hi, mid, lo = split3(x)
    for part in (hi, mid, lo):
        assert not np.any(part.view(np.uint32) & np.uint32(0x0000FFFF))          # bf16-representable: nothing below bit 16
    # exact: the sum in float64 is the number itself (every subtraction above was exact)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # the parts shrink by 2^-8 each (truncation keeps 8 significand bits)
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(mid) <= ax * 2.0 ** -7) and np.all(np.abs(lo) <= ax * 2.0 ** -15)
    assert np.all(np.signbit(mid[mid != 0]) == np.signbit(x[mid != 0]))          # truncation: residuals keep the sign


def test_the_kept_products_are_exact_in_float32_and_the_dropped_ones_below_one_rounding():
    a, b = samples(100000, 1), samples(100000, 2)
    keep = ~(np.isinf(a.astype(np.float64) * b.astype(np.float64)) | (np.abs(a.astype(np.float64) * b.astype(np.float64)) > 3e38))
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
    assert np.all(np.abs(dropped) <= np.abs(exact) * 2.0 ** -21)                  # |mid| < 2^-7 |x|, |lo| < 2^-15 |x|: 2 x 2^-22 + 2^-30
    assert np.sqrt(np.mean((dropped / exact) ** 2)) < 2.0 ** -23.8                 # typical size: one float32 rounding of the product


def rne_bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def test_a_round_to_nearest_split_would_drop_eight_times_less():
    """tools/split_rne_probe.hpp (lab): hi and mid rounded to nearest (v_cvt_pk_bf16_f32) instead of truncated - still exact, the same
    instruction count, and the 3 dropped partial products fall below half a float32 rounding in the worst case."""
    a, b = samples(100000, 1), samples(100000, 2)
    exact = a.astype(np.float64) * b.astype(np.float64)
    keep = (np.abs(exact) < 3e38) & (np.abs(exact) > 1e-30) & (np.abs(a) < 1e38) & (np.abs(b) < 1e38)
    a, b, exact = a[keep], b[keep], exact[keep]

    def split(x):
        hi = rne_bf16(x)
        r1 = x - hi
        mid = rne_bf16(r1)
        return hi, mid, r1 - mid
    ah, am, al = split(a)
    bh, bm, bl = split(b)
    assert np.array_equal(ah.astype(np.float64) + am + al, a.astype(np.float64))
    assert not np.any(al.view(np.uint32) & np.uint32(0x0000FFFF))               # lo needs no rounding: 8 bits are left
    total = sum(p.astype(np.float64) * q.astype(np.float64) for p, q in ((am, bh), (ah, bm), (ah, bh), (al, bh), (ah, bl), (am, bm)))
    rel = (exact - total) / exact
    assert np.max(np.abs(rel)) < 2.0 ** -24 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -27


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
    assert e6 < 2.0 * e1 + 1e-7 and e3 > 3.0 * e6, (e6, e3, e1)
