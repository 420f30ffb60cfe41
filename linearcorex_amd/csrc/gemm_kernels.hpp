// gemm_kernels.hpp - the two X-streaming MFMA kernels of the Linear CorEx fit loop (gfx950).
//
//   gemm_nt :  Ypart[s][n][j] = sum_{v in split s}  X[n][v] * B[v][j]      (reference: x.dot(ws.T),
//              linearcorex.py:247 / :210 / :226;  B is W or grad stored variable-major [V][Mp])
//   gemm_tn :  Dpart[s][v][j] = sum_{n in split s}  A[n][v] * B[n][j]      (reference: x.T.dot(y),
//              linearcorex.py:259 / :211; also reused for the m x m Gram contractions Y^T.Y, W.W^T
//              and H, linearcorex.py:261 / :294)
//
// Design (CDNA4, wave64):
//   * the output is tall and skinny (Mp = padded n_hidden <= 128 columns), so one wave owns ALL Mp
//     columns of its row tile and X is read from HBM exactly once per launch;
//   * X is used by exactly one wave, so it goes global -> VGPR directly (no LDS round trip);
//     each lane issues 16-byte loads and every 16-lane group covers full 128-byte lines;
//   * MFMA 16x16x4 (f32 and f64 share the A/B lane layout: lane l feeds A[l&15][l>>4] and
//     B[l>>4][l&15]).  The contraction index handled by lane-group q=l>>4 at step e is chosen as
//     k = chunk + q*EL + e, so a lane's EL consecutive elements are EL consecutive MFMA steps
//     (any permutation of k is legal as long as A and B agree);
//   * the small operand B lives in [k][Mp] layout; a lane loads CT=Mp/16 consecutive columns
//     j = c*CT+u and uses element u for column tile u - so B loads and the epilogue are wide too
//     (output column permutation, undone when the tile is written);
//   * split over the contraction (KW waves per block + grid.y splits) keeps >= 8 waves per CU in
//     flight for HBM latency hiding; the KW partial tiles are summed through LDS in a fixed order
//     and the grid-level partials are summed by the consumer kernel: deterministic, no atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcx {

template <typename T> struct MF;
template <> struct MF<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4*(lane>>4) + reg
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};
template <> struct MF<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout differs: col = lane & 15, row = (lane>>4) + 4*reg
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};

template <typename T, int N> struct alignas((sizeof(T) * N) < 16 ? (sizeof(T) * N) : 16) Pk {
    T v[N];
};

template <typename T, int N>
__device__ __forceinline__ Pk<T, N> ldg(const T* p) {
    return *reinterpret_cast<const Pk<T, N>*>(p);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt: rows of X (contiguous contraction) times B[k][Mp].
// grid = (row groups of 16*RT, nsplit); block = 64*KW threads.
// X must be padded: rows to a multiple of 16*RT, ldx to a multiple of CH (= 128 B), zero filled;
// B must have ldx rows (zero rows past V).
// ------------------------------------------------------------------------------------------------
template <typename T, int CT, int RT, int KW>
__global__ void __launch_bounds__(64 * KW)
gemm_nt_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ B, T* __restrict__ out,
               int64_t out_rows, int nchunks, int nsplit, const int* __restrict__ skip_flag) {
    constexpr int Mp = 16 * CT;
    constexpr int EL = 32 / (int)sizeof(T);   // elements per lane per chunk (two 16 B loads)
    constexpr int CH = 4 * EL;                // chunk = 128 B of every row
    typedef typename MF<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);  // [KW][16*RT][Mp]

    if (skip_flag != nullptr && *skip_flag != 0) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (16 * RT);
    const int part = blockIdx.y * KW + wave, nparts = nsplit * KW;
    const int c0 = (int)((int64_t)nchunks * part / nparts);
    const int c1 = (int)((int64_t)nchunks * (part + 1) / nparts);

    acc_t acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};

    const T* xp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) xp[t] = X + (row0 + 16 * t + r) * ldx + q * EL;
    const T* bp = B + (int64_t)(q * EL) * Mp + r * CT;

    Pk<T, EL> a0[RT], a1[RT];
    Pk<T, CT> b0[EL], b1[EL];

#define LCX_NT_LOAD(C, AA, BB)                                                        \
    {                                                                                 \
        const int64_t koff = (int64_t)(C) * CH;                                       \
        _Pragma("unroll") for (int t = 0; t < RT; ++t) AA[t] = ldg<T, EL>(xp[t] + koff); \
        _Pragma("unroll") for (int e = 0; e < EL; ++e) BB[e] = ldg<T, CT>(bp + (koff + e) * Mp); \
    }
#define LCX_NT_MMA(AA, BB)                                                            \
    {                                                                                 \
        _Pragma("unroll") for (int e = 0; e < EL; ++e)                                \
        _Pragma("unroll") for (int t = 0; t < RT; ++t)                                \
        _Pragma("unroll") for (int u = 0; u < CT; ++u)                                \
            acc[t][u] = MF<T>::mma(AA[t].v[e], BB[e].v[u], acc[t][u]);                \
    }

    if (c0 < c1) {
        LCX_NT_LOAD(c0, a0, b0);
        int c = c0;
        while (true) {
            int cn = (c + 1 < c1) ? c + 1 : c1 - 1;
            LCX_NT_LOAD(cn, a1, b1);
            LCX_NT_MMA(a0, b0);
            if (++c >= c1) break;
            cn = (c + 1 < c1) ? c + 1 : c1 - 1;
            LCX_NT_LOAD(cn, a0, b0);
            LCX_NT_MMA(a1, b1);
            if (++c >= c1) break;
        }
    }
#undef LCX_NT_LOAD
#undef LCX_NT_MMA

    // ---- reduce the KW partial tiles through LDS in a fixed order and write the tile ----------
    constexpr int TILE = 16 * RT * Mp;
    T* mine = red + wave * TILE;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                mine[(16 * t + MF<T>::row(lane, g)) * Mp + r * CT + u] = acc[t][u][g];
    __syncthreads();
    T* dst = out + ((int64_t)blockIdx.y * out_rows + row0) * Mp;
    for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) {
        T s = red[idx];
#pragma unroll
        for (int w = 1; w < KW; ++w) s += red[w * TILE + idx];
        dst[idx] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: columns of A (contraction over rows) times B[k][Mp].
// grid = (column tiles of 16*RT, nsplit); block = 64*KW threads.
// A: [K][lda] (or panel-major, see below) with K a multiple of 64 and the tile columns in bounds;
// optional per-row scale of A (used for H, linearcorex.py:294).
// ------------------------------------------------------------------------------------------------
// MODE is for ablation probes only (tools/gemm_probe.hip): 0 = real kernel, 1 = loads without MFMA,
// 2 = MFMA without loads (registers loaded once).
template <typename T, int CT, int RT, int KW, bool SCALE, int MODE, int U>
__device__ __forceinline__ void
tn_body(const T* __restrict__ A, int64_t lda, int64_t tile_stride, const T* __restrict__ B,
        const T* __restrict__ rowscale, T* __restrict__ out, int64_t out_rows, int kgroups,
        int nsplit, const int tile_x, const int split_y) {
    constexpr int Mp = 16 * CT;
    // U = MFMA steps per group; a group is 4*U rows of A (kgroups counts 16-row units)
    typedef typename MF<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);  // [KW][16*RT][Mp]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t v0 = (int64_t)tile_x * (16 * RT);
    const int part = split_y * KW + wave, nparts = nsplit * KW;
    const int ng = kgroups * 4 / U;           // groups of 4*U rows (K is a multiple of 64)
    const int g0 = (int)((int64_t)ng * part / nparts);
    const int g1 = (int)((int64_t)ng * (part + 1) / nparts);

    acc_t acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};

    // plain row-major A: tile_stride = 16*RT, lda = row length.  Panel-major A (each column tile
    // stored as its own contiguous [K][16*RT] slab): tile_stride = K*16*RT, lda = 16*RT, so a
    // wave streams one contiguous region of HBM.
    const T* ap = A + (int64_t)tile_x * tile_stride + (int64_t)q * lda + i * RT;
    const T* bp = B + (int64_t)q * Mp + i * CT;

    Pk<T, RT> a0[U], a1[U];
    Pk<T, CT> b0[U], b1[U];
    T s0[U], s1[U];

#define LCX_TN_LOAD(G, AA, BB, SS)                                                    \
    if (MODE != 2 || (G) == g0) {                                                     \
        const int64_t rb = (int64_t)(G) * (4 * U);                                    \
        _Pragma("unroll") for (int st = 0; st < U; ++st) {                            \
            AA[st] = ldg<T, RT>(ap + (rb + 4 * st) * lda);                            \
            BB[st] = ldg<T, CT>(bp + (rb + 4 * st) * Mp);                             \
            if (SCALE) SS[st] = rowscale[rb + 4 * st + q];                            \
        }                                                                             \
    }
#define LCX_TN_MMA(AA, BB, SS)                                                        \
    {                                                                                 \
        _Pragma("unroll") for (int st = 0; st < U; ++st)                              \
        _Pragma("unroll") for (int t = 0; t < RT; ++t) {                              \
            const T av = SCALE ? AA[st].v[t] * SS[st] : AA[st].v[t];                  \
            _Pragma("unroll") for (int u = 0; u < CT; ++u) {                          \
                if (MODE == 1) { asm volatile("" ::"v"(av), "v"(BB[st].v[u])); }      \
                else acc[t][u] = MF<T>::mma(av, BB[st].v[u], acc[t][u]);              \
            }                                                                         \
        }                                                                             \
    }

    if (g0 < g1) {
        LCX_TN_LOAD(g0, a0, b0, s0);
        if (MODE == 2) { LCX_TN_LOAD(g0, a1, b1, s1); }
        int g = g0;
        while (true) {
            int gn = (g + 1 < g1) ? g + 1 : g1 - 1;
            LCX_TN_LOAD(gn, a1, b1, s1);
            LCX_TN_MMA(a0, b0, s0);
            if (++g >= g1) break;
            gn = (g + 1 < g1) ? g + 1 : g1 - 1;
            LCX_TN_LOAD(gn, a0, b0, s0);
            LCX_TN_MMA(a1, b1, s1);
            if (++g >= g1) break;
        }
    }
#undef LCX_TN_LOAD
#undef LCX_TN_MMA

    constexpr int TILE = 16 * RT * Mp;
    T* mine = red + wave * TILE;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                mine[(MF<T>::row(lane, g) * RT + t) * Mp + i * CT + u] = acc[t][u][g];
    __syncthreads();
    T* dst = out + ((int64_t)split_y * out_rows + v0) * Mp;
    for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) {
        T s = red[idx];
#pragma unroll
        for (int w = 1; w < KW; ++w) s += red[w * TILE + idx];
        dst[idx] = s;
    }
}

template <typename T, int CT, int RT, int KW, bool SCALE, int MODE = 0, int U = 4>
__global__ void __launch_bounds__(64 * KW)
gemm_tn_kernel(const T* __restrict__ A, int64_t lda, int64_t tile_stride, const T* __restrict__ B,
               const T* __restrict__ rowscale, T* __restrict__ out, int64_t out_rows, int kgroups,
               int nsplit, const int* __restrict__ skip_flag) {
    if (skip_flag != nullptr && *skip_flag != 0) return;
    tn_body<T, CT, RT, KW, SCALE, MODE, U>(A, lda, tile_stride, B, rowscale, out, out_rows, kgroups, nsplit,
                                            blockIdx.x, blockIdx.y);
}

// Two independent Gram contractions (A^T.A of two [K][Mp] arrays) in one launch: blockIdx.z picks
// the problem, grid.y = max of the two split counts.
template <typename T> struct GramProblem {
    const T* A;
    T* out;
    int kgroups, nsplit;
};
template <typename T, int CT, int RT, int KW>
__global__ void __launch_bounds__(64 * KW)
gram_pair_kernel(GramProblem<T> p0, GramProblem<T> p1) {
    const GramProblem<T> p = blockIdx.z ? p1 : p0;
    if ((int)blockIdx.y >= p.nsplit) return;
    tn_body<T, CT, RT, KW, false, 0, 4>(p.A, 16 * CT, 16 * RT, p.A, nullptr, p.out, 16 * CT, p.kgroups, p.nsplit,
                                         blockIdx.x, blockIdx.y);
}

// tile shapes per (dtype, CT): chosen so accumulators + two register sets stay under ~200 VGPRs
template <typename T, int CT> struct NtShape { static constexpr int RT = (sizeof(T) == 8 && CT >= 8) ? 1 : 2; };
// tn: wave tile = 16*RT columns of A.  Measured on MI355X (tools/gemm_probe, 10k x 5k f64, Mp=32):
// 64-column tiles (512 B per row per wave) stream ~8 % faster than 32-column ones.
template <typename T, int CT> struct TnShape {
    static constexpr int RT = (sizeof(T) == 8) ? (CT >= 8 ? 1 : (CT >= 4 ? 2 : 4)) : (CT >= 8 ? 2 : 4);
};

}  // namespace lcx
