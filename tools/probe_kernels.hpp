// probe_kernels.hpp - kernel variants that only the probes under tools/ instantiate (not part of the library).
#pragma once
#include "../linearcorex_amd/csrc/gemm_kernels.hpp"

namespace lcx {

// ------------------------------------------------------------------------------------------------
// gemm_ct3: gemm_ct with the A operand prefetched TWO groups ahead (a ring of three register sets) and the B chunk
// requested before the A rows of the same group, so that the wait in front of the LDS store only covers loads that are two
// (A) / one (B) MFMA bursts old.  Probe variant (tools/gemm_probe8): the production kernel waits for everything
// (`s_waitcnt vmcnt(0)`) one burst after issuing it, which under a 3.9 TB/s stream is about the loaded-HBM latency.
// ------------------------------------------------------------------------------------------------
template <typename T, int CT, int RT, int KW, int U, bool NT = true>
__global__ void __launch_bounds__(64 * KW)
gemm_ct3_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, T* __restrict__ out,
                int64_t out_rows, int64_t vcols, int ng, int nsuper, int maxslots, const int* __restrict__ skip_flag) {
    constexpr int Mp = 16 * CT;
    constexpr int CHUNK = 4 * U * Mp;
    constexpr int PCS = CHUNK * (int)sizeof(T) / 16;
    constexpr int NTH = 64 * KW;
    constexpr int PPT = (PCS + NTH - 1) / NTH;
    typedef typename MF<T>::acc_t acc_t;
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) T Bs[2][CHUNK];
    if (skip_flag != nullptr && *skip_flag != 0) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
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
        const bool active = v0 < vcols;

        acc_t acc[RT][CT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};
        const T* ap = A + (active ? v0 : 0) + (int64_t)q * lda;
        T a0[U][RT], a1[U][RT], a2[U][RT];
        f4 bst[PPT];

        // loads are unconditional (past the end the last group is fetched again; inactive waves read tile 0): straight-line
        // code lets the compiler count outstanding loads exactly instead of waiting for all of them at every join
#define LCX_C3_LOADA(R, AA)                                                               \
        {                                                                                 \
            const int64_t rb = (int64_t)(s0 + ((R) < cnt ? (R) : cnt - 1)) * (4 * U);     \
            _Pragma("unroll") for (int st = 0; st < U; ++st)                              \
                load_row_pieces<T, RT, NT>(ap + (rb + 4 * st) * lda, i, AA[st]);          \
        }
#define LCX_C3_LOADB(R)                                                                   \
        {                                                                                 \
            const f4* src = reinterpret_cast<const f4*>(B + (int64_t)(s0 + ((R) < cnt ? (R) : cnt - 1)) * CHUNK); \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) bst[p] = src[pc];                         \
            }                                                                             \
        }
#define LCX_C3_STOREB(BUF)                                                                \
        {                                                                                 \
            f4* dstp = reinterpret_cast<f4*>(&Bs[BUF][0]);                                \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) dstp[pc] = bst[p];                        \
            }                                                                             \
        }
#define LCX_C3_MMA(AA, BUF)                                                               \
        if (active) {                                                                     \
            Pk<T, CT> bb[U];                                                              \
            _Pragma("unroll") for (int st = 0; st < U; ++st)                              \
                bb[st] = *reinterpret_cast<const Pk<T, CT>*>(&Bs[BUF][(4 * st + q) * Mp + i * CT]); \
            _Pragma("unroll") for (int st = 0; st < U; ++st)                              \
            _Pragma("unroll") for (int t = 0; t < RT; ++t)                                \
            _Pragma("unroll") for (int u = 0; u < CT; ++u)                                \
                acc[t][u] = MF<T>::mma(AA[st][t], bb[st].v[u], acc[t][u]);                \
        }
        // step r: B chunk r is in bst, A rows r in ring slot r % 3, A rows r+1 already in flight
#define LCX_C3_STEP(CUR, NXT2, BUF)                                                       \
        {                                                                                 \
            LCX_C3_STOREB(BUF);                                                           \
            LCX_C3_LOADB(r + 1);                                                          \
            LCX_C3_LOADA(r + 2, NXT2);                                                    \
            __syncthreads();                                                              \
            LCX_C3_MMA(CUR, BUF);                                                         \
        }

        LCX_C3_LOADB(0);
        LCX_C3_LOADA(0, a0);
        LCX_C3_LOADA(1, a1);
        int r = 0;
        while (true) {
            LCX_C3_STEP(a0, a2, 0); if (++r >= cnt) break;
            LCX_C3_STEP(a1, a0, 1); if (++r >= cnt) break;
            LCX_C3_STEP(a2, a1, 0); if (++r >= cnt) break;
            LCX_C3_STEP(a0, a2, 1); if (++r >= cnt) break;
            LCX_C3_STEP(a1, a0, 0); if (++r >= cnt) break;
            LCX_C3_STEP(a2, a1, 1); if (++r >= cnt) break;
        }
#undef LCX_C3_LOADA
#undef LCX_C3_LOADB
#undef LCX_C3_STOREB
#undef LCX_C3_MMA
#undef LCX_C3_STEP

        const int fb = sk_owner((int64_t)st_ * ng, total, nb);
        if (active) {
            T* dst = out + ((int64_t)(blockIdx.x - fb) * out_rows + v0) * Mp;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    Pk<T, CT> o;
#pragma unroll
                    for (int u = 0; u < CT; ++u) o.v[u] = acc[t][u][g];
                    *reinterpret_cast<Pk<T, CT>*>(dst + piece_col<T, RT>(t, MF<T>::row(lane, g)) * Mp + i * CT) = o;
                }
            if (s1 == ng) {
                const int lb = sk_owner((int64_t)st_ * ng + ng - 1, total, nb);
                Pk<T, CT> z;
#pragma unroll
                for (int u = 0; u < CT; ++u) z.v[u] = (T)0;
                for (int sl = lb - fb + 1; sl < maxslots; ++sl) {
                    T* zd = out + ((int64_t)sl * out_rows + v0) * Mp;
#pragma unroll
                    for (int t = 0; t < RT; ++t)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<Pk<T, CT>*>(zd + (16 * t + 4 * g + q) * Mp + i * CT) = z;
                }
            }
        }
        __syncthreads();
        L0 += cnt;
    }
}

}  // namespace lcx
