// split32_probe.hpp - lab only (tools/gemm_probe9 s32): the X.B^T pass of the bf16-split contraction on the 32x32x16 tile
// (v_mfma_f32_32x32x16_bf16: half the MFMA instructions and operand reads per MAC of the 16x16x32 form production uses).
// Same unit / slot contract as gemm_split_kernel; a wave owns 64 rows = two 32-row tiles; a group is 32 contraction elements = the
// two panels 2G, 2G + 1 = two MFMA steps of 16.  Lane (i = l % 32, kg = l / 32): A = row v0 + 32 rt + i, the 8 floats of chunks
// 2 kg, 2 kg + 1 of the step's panel (two 16-byte loads: a load instruction covers HALF of every 64-byte row segment - the price of
// the 32-row tile); B = column j CT32 + u (so a lane's CT32 accumulators of a row are consecutive columns), contraction rows
// 32 G + 16 ks + 8 kg + e.  D: col = l % 32, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
#pragma once
#include "../linearcorex_amd/csrc/gemm_split_kernels.hpp"

namespace lcx {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16_t mma32_bf16(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Bsp32[group][ks][part][u][lane] x 16 bytes
template <int CT32>
__global__ void __launch_bounds__(256)
split_b32_kernel(const float* __restrict__ B /* [K][Mp] */, u32x4_t* __restrict__ Bsp, int ng) {
    constexpr int Mp = 32 * CT32, TASKS = 2 * 2 * Mp;            // (ks, kg, column)
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < (int64_t)ng * TASKS; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t G = w / TASKS;
        const int k = (int)(w - G * TASKS);
        const int j = k & 31, u = (k >> 5) % CT32, kg = (k / Mp) & 1, ks = k / (2 * Mp);
        const float* src = B + (G * 32 + 16 * ks + 8 * kg) * Mp + j * CT32 + u;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = src[e * Mp];
        const Split3 s = split8(x);
#pragma unroll
        for (int q = 0; q < 3; ++q) Bsp[((G * 2 + ks) * 3 + q) * (CT32 * 64) + u * 64 + kg * 32 + j] = s.p[q];
    }
}

template <int CT32, int KW, int NP, bool NT, int PRIO>
__global__ void __launch_bounds__(64 * KW, 2)
gemm_split32_kernel(const float* __restrict__ A, int64_t ps, const u32x4_t* __restrict__ Bsp, float* __restrict__ out, int64_t out_rows,
                    int64_t nrows, int ng, int nsuper, int maxslots) {
    constexpr int Mp = 32 * CT32, NTH = 64 * KW;
    constexpr int PCS = 2 * 3 * CT32 * 64;
    constexpr int PPT = (PCS + NTH - 1) / NTH;
    __shared__ u32x4_t Bs[2][PCS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, kg = lane >> 5;
    const int64_t total = (int64_t)nsuper * ng;
    const int nb = gridDim.x;
    int64_t L0 = total * blockIdx.x / nb;
    const int64_t L1 = total * (blockIdx.x + 1) / nb;
    while (L0 < L1) {
        const int st_ = (int)(L0 / ng);
        const int s0 = (int)(L0 - (int64_t)st_ * ng);
        const int s1 = (L1 - L0) < (int64_t)(ng - s0) ? s0 + (int)(L1 - L0) : ng;
        const int cnt = s1 - s0;
        const int64_t v0 = ((int64_t)st_ * KW + wave) * 64;
        const bool active = v0 < nrows;
        f32x16_t acc[2][CT32];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < CT32; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
        const float* ap = A + ((active ? v0 : 0) + i) * 16 + kg * 8;
        f32x4_t raw[2][2][2];                                    // [row tile][ks][chunk]
        u32x4_t bst[PPT];
        Split3 as[2][2];
#define S32_LOADA(R)                                                                      \
        {                                                                                 \
            const int64_t G = s0 + ((R) < cnt ? (R) : cnt - 1);                           \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                 \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                              \
            _Pragma("unroll") for (int c = 0; c < 2; ++c) {                               \
                const f32x4_t* src = reinterpret_cast<const f32x4_t*>(ap + (2 * G + ks) * ps + (int64_t)(32 * t) * 16 + 4 * c); \
                raw[t][ks][c] = NT ? __builtin_nontemporal_load(src) : *src;              \
            }                                                                             \
        }
#define S32_LOADB(R)                                                                      \
        {                                                                                 \
            const u32x4_t* src = Bsp + (int64_t)(s0 + ((R) < cnt ? (R) : cnt - 1)) * PCS; \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) bst[p] = src[pc];                         \
            }                                                                             \
        }
        S32_LOADB(0);
        S32_LOADA(0);
        for (int r = 0; r < cnt; ++r) {
            const int buf = r & 1;
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                const int pc = p * NTH + (int)threadIdx.x;
                if (PCS % NTH == 0 || pc < PCS) Bs[buf][pc] = bst[p];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = raw[t][ks][e >> 2][e & 3];
                    as[t][ks] = split8(x);
                }
            S32_LOADB(r + 1);
            S32_LOADA(r + 1);
            __syncthreads();
            if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int u = 0; u < CT32; ++u) {
                    Split3 b;
#pragma unroll
                    for (int q = 0; q < 3; ++q) b.p[q] = Bs[buf][((ks * 3 + q) * CT32 + u) * 64 + lane];
#pragma unroll
                    for (int k = 8 - NP; k < 8; ++k)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t][u] = mma32_bf16(as[t][ks].p[SPLIT_PA[k]], b.p[SPLIT_PB[k]], acc[t][u]);
                }
            if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
        }
#undef S32_LOADA
#undef S32_LOADB
        const int fb = sk_owner((int64_t)st_ * ng, total, nb);
        if (active) {
            float* dst = out + ((int64_t)(blockIdx.x - fb) * out_rows + v0) * Mp;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    Pk<float, CT32> o;
#pragma unroll
                    for (int u = 0; u < CT32; ++u) o.v[u] = acc[t][u][r];
                    const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    *reinterpret_cast<Pk<float, CT32>*>(dst + row * Mp + i * CT32) = o;
                }
            if (s1 == ng) {
                const int lb = sk_owner((int64_t)st_ * ng + ng - 1, total, nb);
                Pk<float, CT32> z;
#pragma unroll
                for (int u = 0; u < CT32; ++u) z.v[u] = 0.f;
                for (int sl = lb - fb + 1; sl < maxslots; ++sl) {
                    float* zd = out + ((int64_t)sl * out_rows + v0) * Mp;
#pragma unroll
                    for (int rr = 0; rr < 32; ++rr) *reinterpret_cast<Pk<float, CT32>*>(zd + (2 * rr + kg) * Mp + i * CT32) = z;
                }
            }
        }
        __syncthreads();
        L0 += cnt;
    }
}

}  // namespace lcx
