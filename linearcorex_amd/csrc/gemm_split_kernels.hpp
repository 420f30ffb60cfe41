// gemm_split_kernels.hpp - the two float32 X-streaming contractions on the bf16 matrix pipe (gfx950), float32 in, float32 out.
//
// Why: on MI355X the float32 MFMA (v_mfma_f32_16x16x4_f32) runs at the float32 VECTOR rate, 1/16 of the bf16 MFMA
// (/opt/skills/guides/MI355X_MICROARCH.md: 157 TF/s against 2.5 PF/s), and the large float32 shards of the fit loop are bound by it
// (gemm_cr / gemm_ct at 0.80-0.92 of that peak, 4-5.5 TB/s of the 8 TB/s HBM).  A float32 number is EXACTLY the sum of three bf16
// numbers (24 significand bits = 8 + 8 + 8), a bf16 x bf16 product is exact in float32, and the matrix pipe accumulates in float32.  So
//
//     a.b = (ah + am + al).(bh + bm + bl) = ah.bh + (ah.bm + am.bh) + (am.bm + ah.bl + al.bh) + [am.bl + al.bm] + {al.bl}
//
// and NP = 6 products drop the [..] pair and {..}.
//
// The split is ROUND-TO-NEAREST (round 5; round 4's truncation split is gone): hi = bf16(x) and mid = bf16(x - hi) by
// v_cvt_pk_bf16_f32 (round to nearest even, two elements per instruction), lo = x - hi - mid; both subtractions are exact and lo has
// at most 8 significant bits, so x = hi + mid + lo exactly (tests/test_split_arithmetic_cpu.py repeats it in NumPy).  The residuals are
// signed and at most half a bf16 ulp: |mid| <= 2^-8 |x|, |lo| <= 2^-17 |x|.  The three dropped products are then at most 2^-24.4
// of the product and 2^-27 of it in the rms, ZERO-MEAN - below half a float32 rounding of the product, which the float32 MFMA chain
// commits at every one of its K steps anyway.  (A truncation split - mask the upper 16 bits - measured 6 % faster at full
// size, but its residuals all carry the sign of x: the dropped terms are a systematic shrink of 2^-24 .. 2^-21.3 of every
// product that adds up coherently in same-sign sums such as uj = sum y^2.  An opt-in arithmetic should not have a bias; measured
// error against a float64 contraction, full-size operands: 1.0-1.2 x the float32 MFMA's with this split, 1.1-1.4 x with truncation,
// profiles/r04_gemm_probe9_rne.txt.)  Caveat: rounding to nearest overflows to infinity for |x| >= 2^128 (1 - 2^-9) = 3.39e38, within
// half a bf16 ulp of FLT_MAX; the operands here are standardised data and weights of order 1.
// NP = 8 keeps the [..] pair as well (no measurable difference in a contraction: what is left is the float32 accumulation).
// One v_mfma_f32_16x16x32_bf16 (~17 cycles) covers 32 contraction elements, for which the float32 MFMA needs 8 instructions of 32
// cycles: 6 products are 2.5 x less matrix-pipe time; 64-column passes become HBM bound (5.2-5.6 TB/s), 128-column passes run the
// bf16 pipe under the power cap - 1.35-1.45 x the float32-MFMA passes either way.
//
// Same machine as gemm_ct / gemm_cr (gemm_kernels.hpp): a block's KW waves own KW adjacent 64-row output tiles and walk the same
// contraction range; the small operand B is staged once per block through LDS, double buffered, one barrier per group of KS x 32
// contraction elements (KS = 2 up to 64 columns); (super tile, group) units are split stream-K style over one round of resident
// blocks; partial tiles go to slots (fixed count, zero-filled by the last contributor, summed in fixed order by the consumers).
// The engine launches 8-wave blocks of 512 rows, one per CU, from 64 columns on (B is re-read once per super tile: its L2 -> fabric
// traffic halves against 4-wave blocks), with the MFMA phase at raised wave priority (engine.hpp, SplitShape).  What is new:
//   * the A operand (X from the PANEL-major copy XP[v / 16][n][16]) goes global -> VGPR as 16-byte loads exactly as in the float32
//     kernels (1 KB contiguous per load instruction for X.B^T, 4 x 256 B for X^T.Y) and is split in registers (4 VALU operations per
//     element + 3 v_perm_b32 per pair) right before use;
//   * B is split ONCE per pass by split_b_kernel into MFMA operand order (1.5 x its bytes; 10-20 us for 25-50 MB), so the staging of
//     the contraction kernel is a plain copy of 16-byte pieces global -> VGPR -> LDS and a wave's operand fetch is one conflict-free
//     ds_read_b128 per (column tile, part).  (Splitting B inside the contraction kernel, per block, measured 3-9 % slower.)
// Any permutation of the 32 contraction elements of a group over (lane group g, element e) is legal as long as A and B agree:
//   X.B^T  (CONTRACT_N = false; contraction over variables): lane (i, g) loads row v0 + 16 t + i, chunk g of panels 2G and 2G + 1:
//          element e = 4 h + c is variable 32 G + 16 h + 4 g + c;
//   X^T.Y  (CONTRACT_N = true; contraction over samples): lane (i, g) loads the 4 variables v0 + 4 i .. + 3 of samples
//          32 G + 4 m + g, m = 0..7 (the 16 lanes i cover 4 panels, the 4 lane groups g 4 consecutive samples = 256 contiguous bytes
//          per panel); element e of row tile t is sample 32 G + 4 e + g of variable v0 + 4 i + t.
#pragma once
#include "gemm_kernels.hpp"

namespace lcx {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Split3 { u32x4_t p[3]; };             // hi, mid, lo

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// 8 floats -> three packed bf16x8 operands (element 2p in the low half of word p): round-to-nearest three-way split, exact
// (hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid; v_cvt_pk_bf16_f32 converts a pair per instruction)
__device__ __forceinline__ Split3 split8(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
        const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xffff0000u);
        const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){ra, rb}, bf16x2_t));
        const float la = ra - __uint_as_float(mp << 16), lb = rb - __uint_as_float(mp & 0xffff0000u);
        s.p[0][p] = hp;
        s.p[1][p] = mp;
        s.p[2][p] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
    return s;
}

__device__ __forceinline__ f32x4_t mma_bf16(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// the products in the order they are accumulated, smallest terms first: (part of A, part of B); NP = 3 / 6 / 8 takes the LAST NP
__device__ constexpr int SPLIT_PA[8] = {1, 2, 2, 0, 1, 1, 0, 0};
__device__ constexpr int SPLIT_PB[8] = {2, 1, 0, 2, 1, 0, 1, 0};

constexpr int SPLIT_KG = 32;                 // contraction elements per group (one bf16 MFMA step)

// ------------------------------------------------------------------------------------------------
// split_b_kernel: the small operand B[K][Mp] split once per pass, in MFMA operand order:
// Bsp[group][part][column tile u][lane] x 16 bytes - the LDS image of a group, contiguous (1.5 x the bytes of B).
// ------------------------------------------------------------------------------------------------
template <int CT, bool CONTRACT_N>
__global__ void __launch_bounds__(256)
split_b_kernel(const float* __restrict__ B /* [K][Mp] */, u32x4_t* __restrict__ Bsp, int ng, const int* __restrict__ skip_flag) {
    constexpr int Mp = 16 * CT, TASKS = 4 * Mp;
    if (skip_flag != nullptr && *skip_flag != 0) return;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < (int64_t)ng * TASKS; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t G = w / TASKS;
        const int k = (int)(w - G * TASKS);
        const int j = k & 15, u = (k >> 4) % CT, gg = k / (16 * CT);
        const float* src = B + G * SPLIT_KG * Mp;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = CONTRACT_N ? 4 * e + gg : 16 * (e >> 2) + 4 * gg + (e & 3);
            x[e] = src[row * Mp + j * CT + u];
        }
        const Split3 s = split8(x);
#pragma unroll
        for (int q = 0; q < 3; ++q) Bsp[(G * 3 + q) * (CT * 64) + u * 64 + gg * 16 + j] = s.p[q];
    }
}

template <int CT, int KW, int NP, bool CONTRACT_N, bool NT, bool PREFETCH_B, int WPE = 2, int KS = 1, int PRIO = 0>
__global__ void __launch_bounds__(64 * KW, WPE)
gemm_split_kernel(const float* __restrict__ A /* panel-major */, int64_t ps, const u32x4_t* __restrict__ Bsp, float* __restrict__ out, int64_t out_rows,
                   int64_t nrows, int ng /* groups of KS x 32 */, int nsuper, int maxslots, const int* __restrict__ skip_flag) {
    constexpr int Mp = 16 * CT, RT = 4, NTH = 64 * KW;
    constexpr int PC1 = 3 * CT * 64;                         // 16-byte pieces of 32 contraction elements of B
    constexpr int PCS = KS * PC1;                            // ... of one group
    constexpr int PPT = (PCS + NTH - 1) / NTH;
    __shared__ u32x4_t Bs[2][PCS];
    if (skip_flag != nullptr && *skip_flag != 0) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int64_t total = (int64_t)nsuper * ng;
    const int nb = gridDim.x;
    int64_t L0 = total * blockIdx.x / nb;
    const int64_t L1 = total * (blockIdx.x + 1) / nb;

    while (L0 < L1) {
        const int st_ = (int)(L0 / ng);
        const int s0 = (int)(L0 - (int64_t)st_ * ng);
        const int s1 = (L1 - L0) < (int64_t)(ng - s0) ? s0 + (int)(L1 - L0) : ng;
        const int cnt = s1 - s0;
        const int64_t v0 = ((int64_t)st_ * KW + wave) * (16 * RT);
        const bool active = v0 < nrows;

        f32x4_t acc[RT][CT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < CT; ++u) acc[t][u] = (f32x4_t){0, 0, 0, 0};

        const float* ap = CONTRACT_N ? A + ((active ? v0 : 0) / 16 + (i >> 2)) * ps + (int64_t)g * 16 + (i & 3) * 4
                                     : A + ((active ? v0 : 0) + i) * 16 + g * 4;
        f32x4_t raw[KS][8];
        u32x4_t bst[PPT];
        Split3 as[RT];

        // the 8 x 16 bytes of k-step S of group R (clamped to the segment: the tail re-loads its last group instead of branching)
#define LCX_SP_LOADA(R, S)                                                                \
        {                                                                                 \
            const int64_t G = (int64_t)(s0 + ((R) < cnt ? (R) : cnt - 1)) * KS + (S);     \
            _Pragma("unroll") for (int m = 0; m < 8; ++m) {                               \
                const f32x4_t* src = CONTRACT_N ? reinterpret_cast<const f32x4_t*>(ap + (32 * G + 4 * m) * 16) \
                                                : reinterpret_cast<const f32x4_t*>(ap + (2 * G + (m & 1)) * ps + (int64_t)(16 * (m >> 1)) * 16); \
                raw[S][m] = NT ? __builtin_nontemporal_load(src) : *src;                  \
            }                                                                             \
        }
#define LCX_SP_LOADB(R)                                                                   \
        {                                                                                 \
            const u32x4_t* src = Bsp + (int64_t)(s0 + ((R) < cnt ? (R) : cnt - 1)) * PCS; \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) bst[p] = src[pc];                         \
            }                                                                             \
        }
#define LCX_SP_STOREB(BUF)                                                                \
        {                                                                                 \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) Bs[BUF][pc] = bst[p];                     \
            }                                                                             \
        }
#define LCX_SP_SPLITA(S)                                                                  \
        {                                                                                 \
            _Pragma("unroll") for (int t = 0; t < RT; ++t) {                              \
                float x[8];                                                               \
                _Pragma("unroll") for (int e = 0; e < 8; ++e)                             \
                    x[e] = CONTRACT_N ? raw[S][e][t] : raw[S][2 * t + (e >> 2)][e & 3];   \
                as[t] = split8(x);                                                        \
            }                                                                             \
        }
#define LCX_SP_MMA(BUF, S)                                                                \
        if constexpr (PREFETCH_B) {                                                       \
            Split3 bf[2];                                                                 \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) bf[0].p[q] = Bs[BUF][(S) * PC1 + (q * CT) * 64 + lane]; \
            _Pragma("unroll") for (int u = 0; u < CT; ++u) {                              \
                if (u + 1 < CT) {                                                         \
                    _Pragma("unroll") for (int q = 0; q < 3; ++q) bf[(u + 1) & 1].p[q] = Bs[BUF][(S) * PC1 + (q * CT + u + 1) * 64 + lane]; \
                }                                                                         \
                _Pragma("unroll") for (int k = 8 - NP; k < 8; ++k)                        \
                _Pragma("unroll") for (int t = 0; t < RT; ++t)                            \
                    acc[t][u] = mma_bf16(as[t].p[SPLIT_PA[k]], bf[u & 1].p[SPLIT_PB[k]], acc[t][u]); \
            }                                                                             \
        } else {                                                                          \
            _Pragma("unroll") for (int u = 0; u < CT; ++u) {                              \
                Split3 b;                                                                 \
                _Pragma("unroll") for (int q = 0; q < 3; ++q) b.p[q] = Bs[BUF][(S) * PC1 + (q * CT + u) * 64 + lane]; \
                _Pragma("unroll") for (int k = 8 - NP; k < 8; ++k)                        \
                _Pragma("unroll") for (int t = 0; t < RT; ++t)                            \
                    acc[t][u] = mma_bf16(as[t].p[SPLIT_PA[k]], b.p[SPLIT_PB[k]], acc[t][u]); \
            }                                                                             \
        }

        // one barrier per group; the loads of the next group's k-step S are issued as soon as this group's k-step S has been split
        // (B first: the in-order load counter then lets the staging wait on B without waiting on the younger A loads)
        LCX_SP_LOADB(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) LCX_SP_LOADA(0, ks);
        for (int r = 0; r < cnt; ++r) {
            const int buf = r & 1;
            LCX_SP_STOREB(buf);
            LCX_SP_SPLITA(0);
            LCX_SP_LOADB(r + 1);
            LCX_SP_LOADA(r + 1, 0);
            __syncthreads();
            if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
            LCX_SP_MMA(buf, 0);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) {
                if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
                LCX_SP_SPLITA(ks);
                LCX_SP_LOADA(r + 1, ks);
                if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
                LCX_SP_MMA(buf, ks);
            }
            if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
        }
#undef LCX_SP_LOADA
#undef LCX_SP_LOADB
#undef LCX_SP_STOREB
#undef LCX_SP_SPLITA
#undef LCX_SP_MMA

        const int fb = sk_owner((int64_t)st_ * ng, total, nb);
        if (active) {
            float* dst = out + ((int64_t)(blockIdx.x - fb) * out_rows + v0) * Mp;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Pk<float, CT> o;
#pragma unroll
                    for (int u = 0; u < CT; ++u) o.v[u] = acc[t][u][r];
                    const int row = CONTRACT_N ? 4 * (4 * g + r) + t : 16 * t + 4 * g + r;
                    *reinterpret_cast<Pk<float, CT>*>(dst + row * Mp + i * CT) = o;
                }
            if (s1 == ng) {
                const int lb = sk_owner((int64_t)st_ * ng + ng - 1, total, nb);
                Pk<float, CT> z;
#pragma unroll
                for (int u = 0; u < CT; ++u) z.v[u] = 0.f;
                for (int sl = lb - fb + 1; sl < maxslots; ++sl) {
                    float* zd = out + ((int64_t)sl * out_rows + v0) * Mp;
#pragma unroll
                    for (int t = 0; t < RT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            *reinterpret_cast<Pk<float, CT>*>(zd + (16 * t + 4 * r + g) * Mp + i * CT) = z;
                }
            }
        }
        __syncthreads();
        L0 += cnt;
    }
}

}  // namespace lcx
