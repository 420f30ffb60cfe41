// split_rne_probe.hpp - lab only (tools/gemm_probe9 rne): the production split contraction with a ROUND-TO-NEAREST three-way split
// (v_cvt_pk_bf16_f32 for hi and mid, the residuals by exact float32 subtractions) instead of the truncation split: the same 11
// instructions per pair of elements, residuals half the size and signed, so the 3 dropped partial products are at most 2^-24.4 of a
// product (2^-27 rms) instead of 2^-21.3 (2^-24 rms) - tests/test_split_arithmetic_cpu.py.  Caveat: rounding to nearest overflows
// to infinity within half a bf16 ulp of FLT_MAX; the truncation split never does.  Generated from the production kernel text.
#pragma once
#include "../linearcorex_amd/csrc/gemm_split_kernels.hpp"

namespace lcx {

typedef float lab_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 lab_b2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ Split3 split8_rne(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((lab_f2){a, b}, lab_b2));
        const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xffff0000u);
        const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector((lab_f2){ra, rb}, lab_b2));
        const float la = ra - __uint_as_float(mp << 16), lb = rb - __uint_as_float(mp & 0xffff0000u);
        s.p[0][p] = hp;
        s.p[1][p] = mp;
        s.p[2][p] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
    return s;
}

// ------------------------------------------------------------------------------------------------
// split_b_kernel: the small operand B[K][Mp] split once per pass, in MFMA operand order:
// Bsp[group][part][column tile u][lane] x 16 bytes - the LDS image of a group, contiguous (1.5 x the bytes of B).
// ------------------------------------------------------------------------------------------------
template <int CT, bool CONTRACT_N>
__global__ void __launch_bounds__(256)
split_b_rne_kernel(const float* __restrict__ B /* [K][Mp] */, u32x4_t* __restrict__ Bsp, int ng, const int* __restrict__ skip_flag) {
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
        const Split3 s = split8_rne(x);
#pragma unroll
        for (int q = 0; q < 3; ++q) Bsp[(G * 3 + q) * (CT * 64) + u * 64 + gg * 16 + j] = s.p[q];
    }
}

template <int CT, int KW, int NP, bool CONTRACT_N, bool NT, bool PREFETCH_B, int WPE = 2, int KS = 1, int PRIO = 0>
__global__ void __launch_bounds__(64 * KW, WPE)
gemm_split_rne_kernel(const float* __restrict__ A /* panel-major */, int64_t ps, const u32x4_t* __restrict__ Bsp, float* __restrict__ out, int64_t out_rows,
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
                as[t] = split8_rne(x);                                                        \
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
