// split_pp_probe.hpp - lab only (tools/gemm_probe9): gemm_split_kernel (gemm_split_kernels.hpp) with a producer / consumer schedule:
// the 8 waves of a block form two groups that run half an iteration apart, so that on every SIMD one wave is in its MFMA phase while
// the other splits its next A tile and stages B.  Generated from the production kernel text; same arguments and slot contract.
#pragma once
#include "../linearcorex_amd/csrc/gemm_split_kernels.hpp"

namespace lcx {

template <int CT, int NP, bool CONTRACT_N, bool NT, int PRIO>
__global__ void __launch_bounds__(512, 2)
gemm_split_pp_kernel(const float* __restrict__ A /* panel-major */, int64_t ps, const u32x4_t* __restrict__ Bsp, float* __restrict__ out, int64_t out_rows,
                   int64_t nrows, int ng /* groups of KS x 32 */, int nsuper, int maxslots, const int* __restrict__ skip_flag) {
    constexpr int KW = 8, KS = 1; constexpr bool PREFETCH_B = false;
    constexpr int Mp = 16 * CT, RT = 4, NTH = 256;            // NTH: the threads of one wave GROUP (waves 0-3 / 4-7)
    constexpr int PC1 = 3 * CT * 64;                         // 16-byte pieces of 32 contraction elements of B
    constexpr int PCS = KS * PC1;                            // ... of one group
    constexpr int HALF = PCS / 2;                            // each wave group stages one half of a group of B
    constexpr int PPT = (HALF + NTH - 1) / NTH;
    __shared__ u32x4_t Bs[2][PCS];
    if (skip_flag != nullptr && *skip_flag != 0) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave >> 2, tg = threadIdx.x & 255;           // wave group 0: one half-iteration ahead of group 1
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
            const u32x4_t* src = Bsp + (int64_t)(s0 + ((R) < cnt ? (R) : cnt - 1)) * PCS + grp * HALF; \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + tg;                                              \
                if (HALF % NTH == 0 || pc < HALF) bst[p] = src[pc];                       \
            }                                                                             \
        }
#define LCX_SP_STOREB(BUF)                                                                \
        {                                                                                 \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + tg;                                              \
                if (HALF % NTH == 0 || pc < HALF) Bs[BUF][grp * HALF + pc] = bst[p];      \
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

        // ping-pong: group 0 splits / stages while group 1 multiplies and vice versa - two barriers per iteration, every wave in the
        // VALU phase between one pair and in the MFMA phase between the next; group 1 runs half an iteration behind.  B(r): group 0's
        // half is staged in its VALU phase of iteration r, group 1's half one phase earlier (its VALU phase of iteration r - 1)
        LCX_SP_LOADB(0);
        LCX_SP_LOADA(0, 0);
        if (grp == 1) {
            LCX_SP_STOREB(0);
            LCX_SP_LOADB(1);
            __syncthreads();
        }
        for (int r = 0; r < cnt; ++r) {
            const int buf = r & 1;
            if (grp == 0) { LCX_SP_STOREB(buf); } else { LCX_SP_STOREB(buf ^ 1); }
            LCX_SP_SPLITA(0);
            if (grp == 0) { LCX_SP_LOADB(r + 1); } else { LCX_SP_LOADB(r + 2); }
            LCX_SP_LOADA(r + 1, 0);
            __syncthreads();
            if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
            LCX_SP_MMA(buf, 0);
            if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
        if (grp == 0) __syncthreads();
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
