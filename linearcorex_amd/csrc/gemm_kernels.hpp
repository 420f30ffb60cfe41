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
// the same load with the non-temporal hint (streamed-once operand), in 16-byte pieces where the size allows
template <typename T, int N>
__device__ __forceinline__ Pk<T, N> ldg_nt(const T* p) {
    constexpr int BYTES = (int)sizeof(T) * N;
    Pk<T, N> r;
    if constexpr (BYTES % 16 == 0) {
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const f4_t* src = reinterpret_cast<const f4_t*>(p);
        f4_t* dst = reinterpret_cast<f4_t*>(&r);
#pragma unroll
        for (int k = 0; k < BYTES / 16; ++k) dst[k] = __builtin_nontemporal_load(src + k);
    } else if constexpr (BYTES == 8) {
        typedef float f2_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f2_t*>(&r) = __builtin_nontemporal_load(reinterpret_cast<const f2_t*>(p));
    } else {
        r = *reinterpret_cast<const Pk<T, N>*>(p);
    }
    return r;
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
// WT: the tile is stored write-through (agent-scope relaxed stores), for a consumer on another XCD inside the same launch
template <typename T, int CT, int RT, int KW, bool SCALE, int U, bool NTA = false, bool WT = false>
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
    {                                                                                 \
        const int64_t rb = (int64_t)(G) * (4 * U);                                    \
        _Pragma("unroll") for (int st = 0; st < U; ++st) {                            \
            AA[st] = NTA ? ldg_nt<T, RT>(ap + (rb + 4 * st) * lda) : ldg<T, RT>(ap + (rb + 4 * st) * lda); \
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
                acc[t][u] = MF<T>::mma(av, BB[st].v[u], acc[t][u]);                   \
            }                                                                         \
        }                                                                             \
    }

    if (g0 < g1) {
        LCX_TN_LOAD(g0, a0, b0, s0);
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
        if constexpr (WT) __hip_atomic_store(&dst[idx], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dst[idx] = s;
    }
}

template <typename T, int CT, int RT, int KW, bool SCALE, int U = 4, bool NTA = false>
__global__ void __launch_bounds__(64 * KW)
gemm_tn_kernel(const T* __restrict__ A, int64_t lda, int64_t tile_stride, const T* __restrict__ B,
               const T* __restrict__ rowscale, T* __restrict__ out, int64_t out_rows, int kgroups,
               int nsplit, const int* __restrict__ skip_flag) {
    if (skip_flag != nullptr && *skip_flag != 0) return;
    tn_body<T, CT, RT, KW, SCALE, U, NTA>(A, lda, tile_stride, B, rowscale, out, out_rows, kgroups, nsplit,
                                                 blockIdx.x, blockIdx.y);
}

// ---- in-launch chunk signalling (the pipelined Y exchange, LCX_Y_PIPELINE=signal: DESIGN.md section 6) -----------------------
// ONE launch of a wave-split pass that tells a second stream, row chunk by row chunk, when every partial tile of that chunk has been
// written - so that the chunk's slot reduction and all-reduce run there while the pass goes on, without cutting the pass into
// launches that cannot fill the chip.  The 1-D grid walks (row tile, slot) with the slot fastest: the blocks of a row tile are
// dispatched together and the tiles in chunk order.  A block stores its partial tile WRITE-THROUGH (agent-scope relaxed stores: the
// consumer is a kernel of another stream, started while this launch is still running - whatever sits dirty in an XCD's L2 it would not
// see), waits for the stores to be acknowledged, and one lane draws a ticket of the tile's chunk; the block that draws the last one
// stores the launch's epoch into the chunk's signal word (system scope: the command processor of the waiting stream reads it).
// No cache-wide fence anywhere: a first version handed tiles from block to block inside the launch with agent-scope release /
// acquire fences - every one writes the XCD's L2 back or invalidates it under the other blocks, which keep their B operand there:
// 75 -> 95 us per pass at config 2.  The slot reduction itself is the SAME kernel as without the pipeline (same bits, any slot
// count), launched per chunk behind the signal.  Counters return to zero by themselves; epochs only grow.
constexpr int SIG_MAX_CHUNKS = 16;
struct ChunkSig {
    unsigned int* chunk_cnt;                 // [chunks] partial tiles written
    unsigned int* flag[SIG_MAX_CHUNKS];      // one signal word per chunk (hipMallocSignalMemory)
    int tile_begin[SIG_MAX_CHUNKS + 1];      // chunk c = row tiles [tile_begin[c], tile_begin[c + 1])
    int nchunks;
    unsigned int epoch;
};

__device__ __forceinline__ void chunk_signal_tail(const ChunkSig& sg, int tile, int nslots) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this block's tile is in memory
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        while (c + 1 < sg.nchunks && tile >= sg.tile_begin[c + 1]) ++c;
        const unsigned int want = (unsigned int)(sg.tile_begin[c + 1] - sg.tile_begin[c]) * (unsigned int)nslots;
        const unsigned int t = __hip_atomic_fetch_add(&sg.chunk_cnt[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == want - 1) {
            __hip_atomic_store(&sg.chunk_cnt[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __hip_atomic_store(sg.flag[c], sg.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T, int CT, int RT, int KW, int U = 4>
__global__ void __launch_bounds__(64 * KW)
gemm_tn_sig_kernel(const T* __restrict__ A, int64_t lda, int64_t tile_stride, const T* __restrict__ B, T* __restrict__ out,
                   int64_t out_rows, int kgroups, int nsplit, ChunkSig sg) {
    const int tile = (int)(blockIdx.x / (unsigned)nsplit), slot = (int)(blockIdx.x % (unsigned)nsplit);
    tn_body<T, CT, RT, KW, false, U, false, true>(A, lda, tile_stride, B, nullptr, out, out_rows, kgroups, nsplit, tile, slot);
    chunk_signal_tail(sg, tile, nsplit);
}

// a stand-in for a stream wait-value where the runtime has none: one lane polls the chunk's signal word (bounded: ~4 s, then the
// error word is set and the stream goes on - the caller checks it)
static __global__ void poll_signal_kernel(const unsigned int* flag, unsigned int epoch, unsigned int* err) {
    long spins = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1L << 24)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}
static __global__ void init_signal_kernel(unsigned int* flag, unsigned int value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

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
    tn_body<T, CT, RT, KW, false, 4>(p.A, 16 * CT, 16 * RT, p.A, nullptr, p.out, 16 * CT, p.kgroups, p.nsplit,
                                         blockIdx.x, blockIdx.y);
}


// ------------------------------------------------------------------------------------------------
// gemm_ct: the same contraction D[v][j] = sum_n A[n][v] B[n][j] for MANY column tiles (large shards).
//
// Measured on MI355X (tools/gemm_probe2/4): when every wave fetches its own rows of B from L2 the
// B stream is Mp/(16*RT) times the A stream in bytes (1x at n_hidden 64, 4x at 128 in float32) and
// the kernel stalls at ~80 TF/s; with B staged once per block through LDS it reaches 115-125 TF/s.
//   * a block's KW waves own KW ADJACENT column tiles (a "super tile" of KW*16*RT columns) and walk
//     the SAME contraction range, so one copy of B serves the block: global -> VGPR -> LDS, double
//     buffered, one barrier per group of 4*U rows;
//   * A still goes global -> VGPR directly, 16 bytes per lane, contiguous 256 B per 16-lane row;
//   * the contraction split moves to the grid, balanced stream-K style: the work is the list of
//     (super tile, group) units in super-tile-major order, every block takes the same number of
//     units (+-1), so a launch is one full round of resident blocks whatever the shape;
//   * a block that covers only part of a super tile writes a partial tile into slot
//     (block - first block of that super tile); the last contributor zero-fills the unused slots, so
//     consumers sum a fixed number of slots (`maxslots`) in a fixed order: deterministic, no atomics.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int sk_owner(int64_t unit, int64_t total, int nb) {
    return (int)(((unit + 1) * nb - 1) / total);      // the block b with start(b) <= unit < start(b+1), start(b) = total*b/nb
}

template <typename T, int N> struct VecT;
template <> struct VecT<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecT<double, 1> { typedef double type; };
template <> struct VecT<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecT<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecT<float, 1> { typedef float type; };

// ------------------------------------------------------------------------------------------------
// PANEL-MAJOR layout of the resident shard (round 4): XP[v / PW][n][PW], PW = 64 bytes of one row (16 floats / 8 doubles) - a panel is
// ALL rows of PW consecutive variables, contiguous; panel stride = n_padded * PW elements.  ONE copy serves both contractions with
// whole, contiguous cache lines (tools/gemm_probe4 panel, profiles/r04_gemm_probe4_panel.txt: both passes at or above the speed of
// the two-copy layout, which this replaces for large shards):
//   * X.B^T (contraction over v; gemm_cr<.., PANEL>): lane (i, q) loads 16 bytes of row i at chunk q of ONE panel, so the 16 rows of a
//     load instruction are 16 x 64 B = 1 KB CONTIGUOUS (on the row-major X they are 16 segments of 64 B a row length apart - the
//     4-6 % of gemm_cr), a wave's 4 row tiles 4 KB, a block's 256 rows 16 KB; a group of 4 U contraction elements = U / E panels;
//   * X^T.Y (contraction over n; gemm_ct<.., PANEL>): lane (i, q) loads 16 bytes = consecutive v of row 4 st + q; the 16 lanes i cover
//     256 B / 64 B = 4 panels with 4 rows x 64 B = 256 contiguous bytes in each - the same 4 x 256 B per instruction as on the
//     row-major X - and the 16 rows of a group make 1 KB contiguous per panel.
// ------------------------------------------------------------------------------------------------
template <typename T> struct PanelW { static constexpr int v = 64 / (int)sizeof(T); };

// elements per 16-byte piece (capped by the tile count)
template <typename T, int RT> struct Epl { static constexpr int v = (16 / (int)sizeof(T)) < RT ? (16 / (int)sizeof(T)) : RT; };

// RT elements of one row for lane i: piece p sits at column p*16*EPL + i*EPL, so that every load
// instruction covers 16 lanes x 16 B = 256 contiguous bytes of the row
template <typename T, int RT, bool NT = false>
__device__ __forceinline__ void load_row_pieces(const T* rowp, int i, T (&dst)[RT]) {
    constexpr int EPL = Epl<T, RT>::v;
    typedef typename VecT<T, EPL>::type V;
#pragma unroll
    for (int p = 0; p < RT / EPL; ++p) {
        const V* src = reinterpret_cast<const V*>(rowp + p * 16 * EPL + i * EPL);
        const V v = NT ? __builtin_nontemporal_load(src) : *src;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            if constexpr (EPL == 1) dst[p] = v; else dst[p * EPL + e] = v[e];
        }
    }
}
// column (within the wave's 16*RT tile) that accumulator (tile t, MFMA output row r) belongs to
template <typename T, int RT>
__device__ __forceinline__ int piece_col(int t, int r) {
    constexpr int EPL = Epl<T, RT>::v;
    return (t / EPL) * 16 * EPL + r * EPL + (t % EPL);
}

template <typename T, int CT, int RT, int KW, int U, bool NT = false, bool PANEL = false>
__global__ void __launch_bounds__(64 * KW)
gemm_ct_kernel(const T* __restrict__ A, int64_t lda /* PANEL: the panel stride */, const T* __restrict__ B, T* __restrict__ out,
               int64_t out_rows, int64_t vcols, int ng /* groups of 4*U rows */, int nsuper, int maxslots,
               const int* __restrict__ skip_flag) {
    constexpr int Mp = 16 * CT;
    constexpr int CHUNK = 4 * U * Mp;                        // elements of B per group
    constexpr int PCS = CHUNK * (int)sizeof(T) / 16;         // 16-byte pieces per group
    constexpr int NTH = 64 * KW;
    constexpr int PPT = (PCS + NTH - 1) / NTH;               // pieces per thread
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
        const int st_ = (int)(L0 / ng);                      // super tile
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
        // row-major A: row stride lda.  PANEL: piece p of this lane = columns v0 + p 16 EPL + i EPL .., i.e. panel (col / PW) at
        // offset col % PW; rows are PW elements apart inside a panel
        constexpr int EPLC = Epl<T, RT>::v;
        constexpr int PWC = PanelW<T>::v;
        const T* ap = A + (active ? v0 : 0) + (int64_t)q * lda;
        const T* app[RT / EPLC];
        if constexpr (PANEL) {
#pragma unroll
            for (int p = 0; p < RT / EPLC; ++p) {
                const int64_t col = (active ? v0 : 0) + p * 16 * EPLC + i * EPLC;
                app[p] = A + (col / PWC) * lda + (col % PWC) + (int64_t)q * PWC;
            }
        }
        T a0[U][RT], a1[U][RT];
        f4 bst[PPT];

#define LCX_CT_LOADA(R, AA)                                                               \
        if (active) {                                                                     \
            const int64_t rb = (int64_t)(s0 + (R)) * (4 * U);                             \
            if constexpr (PANEL) {                                                        \
                typedef typename VecT<T, EPLC>::type VP;                                  \
                _Pragma("unroll") for (int st = 0; st < U; ++st)                          \
                _Pragma("unroll") for (int p = 0; p < RT / EPLC; ++p) {                   \
                    const VP* src = reinterpret_cast<const VP*>(app[p] + (rb + 4 * st) * PWC); \
                    const VP v = NT ? __builtin_nontemporal_load(src) : *src;             \
                    _Pragma("unroll") for (int e = 0; e < EPLC; ++e) {                    \
                        if constexpr (EPLC == 1) AA[st][p] = v; else AA[st][p * EPLC + e] = v[e]; \
                    }                                                                     \
                }                                                                         \
            } else {                                                                      \
                _Pragma("unroll") for (int st = 0; st < U; ++st)                          \
                    load_row_pieces<T, RT, NT>(ap + (rb + 4 * st) * lda, i, AA[st]);      \
            }                                                                             \
        }
#define LCX_CT_LOADB(R)                                                                   \
        {                                                                                 \
            const f4* src = reinterpret_cast<const f4*>(B + (int64_t)(s0 + (R)) * CHUNK); \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) bst[p] = src[pc];                         \
            }                                                                             \
        }
#define LCX_CT_STOREB(BUF)                                                                \
        {                                                                                 \
            f4* dstp = reinterpret_cast<f4*>(&Bs[BUF][0]);                                \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) dstp[pc] = bst[p];                        \
            }                                                                             \
        }
#define LCX_CT_MMA(AA, BUF)                                                               \
        if (active) {                                                                     \
            Pk<T, CT> bb[U];                                                              \
            _Pragma("unroll") for (int st = 0; st < U; ++st)                              \
                bb[st] = *reinterpret_cast<const Pk<T, CT>*>(&Bs[BUF][(4 * st + q) * Mp + i * CT]); \
            _Pragma("unroll") for (int st = 0; st < U; ++st)                              \
            _Pragma("unroll") for (int t = 0; t < RT; ++t)                                \
            _Pragma("unroll") for (int u = 0; u < CT; ++u)                                \
                acc[t][u] = MF<T>::mma(AA[st][t], bb[st].v[u], acc[t][u]);                \
        }

        LCX_CT_LOADA(0, a0);
        LCX_CT_LOADB(0);
        int r = 0;
        while (true) {
            LCX_CT_STOREB(0);
            if (r + 1 < cnt) { LCX_CT_LOADA(r + 1, a1); LCX_CT_LOADB(r + 1); }
            __syncthreads();
            LCX_CT_MMA(a0, 0);
            if (++r >= cnt) break;
            LCX_CT_STOREB(1);
            if (r + 1 < cnt) { LCX_CT_LOADA(r + 1, a0); LCX_CT_LOADB(r + 1); }
            __syncthreads();
            LCX_CT_MMA(a1, 1);
            if (++r >= cnt) break;
        }
#undef LCX_CT_LOADA
#undef LCX_CT_LOADB
#undef LCX_CT_STOREB
#undef LCX_CT_MMA

        // ---- the wave's tile goes straight from the accumulators to its slot ---------------------
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
            if (s1 == ng) {        // last contributor of this super tile: zero the slots nobody writes
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
        __syncthreads();            // Bs is reused by the next segment
        L0 += cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_cr: D[r][j] = sum_k A[r][k] B[k][j] with A ROW-major and the contraction along its contiguous axis - X.B^T read from
// X itself, so that a large shard does not need a transposed copy (half the resident bytes).
//
// Same machine as gemm_ct: a block's KW waves own KW adjacent tiles of 16*RT output rows and walk the same contraction
// range, B is staged once per block through LDS (global -> VGPR -> LDS, double buffered, one barrier per group of 4*U
// contraction steps), the (super tile, group) units are split stream-K style over one round of resident blocks, partial
// tiles go to slots with the same contract (fixed slot count, zero-filled by the last contributor).  What differs is the A
// operand: lane (i, q) loads E = 16 / sizeof(T) CONSECUTIVE contraction elements of row i (one 16-byte load; the 4 lane
// groups q cover 64 contiguous bytes of the row), i.e. load p of a group holds k = p*4E + q*E + e.  An MFMA step may take its
// 4 contraction elements in any order as long as both operands agree, so step (p, e) reads the B rows p*4E + q*E + e from LDS.
// ------------------------------------------------------------------------------------------------
template <typename T, int CT, int RT, int KW, int U, bool NT = false, bool PANEL = false>
__global__ void __launch_bounds__(64 * KW)
gemm_cr_kernel(const T* __restrict__ A, int64_t lda /* PANEL: the panel stride */, const T* __restrict__ B, T* __restrict__ out,
               int64_t out_rows, int64_t nrows, int ng /* groups of 4*U contraction elements */, int nsuper, int maxslots,
               const int* __restrict__ skip_flag) {
    constexpr int Mp = 16 * CT;
    constexpr int E = 16 / (int)sizeof(T);                   // contraction elements per 16-byte load
    constexpr int NL = U / E;                                // loads per row tile per group
    static_assert(U % E == 0, "gemm_cr: a group must be whole 16-byte loads");
    constexpr int CHUNK = 4 * U * Mp;
    constexpr int PCS = CHUNK * (int)sizeof(T) / 16;
    constexpr int NTH = 64 * KW;
    constexpr int PPT = (PCS + NTH - 1) / NTH;
    typedef typename MF<T>::acc_t acc_t;
    typedef typename VecT<T, E>::type AV;
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
        const bool active = v0 < nrows;

        acc_t acc[RT][CT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};
        // row-major A: row i at stride lda, the group's elements along the row.  PANEL: row i at stride PW inside panel
        // (group * NL + p), panels `lda` elements apart
        constexpr int PWC = 4 * E;
        const T* ap = PANEL ? A + ((active ? v0 : 0) + i) * PWC + q * E : A + ((active ? v0 : 0) + i) * lda + q * E;
        AV a0[NL][RT], a1[NL][RT];
        f4 bst[PPT];

#define LCX_CR_LOADA(R, AA)                                                               \
        if (active) {                                                                     \
            const int64_t kb = (int64_t)(s0 + (R)) * (4 * U);                             \
            _Pragma("unroll") for (int p = 0; p < NL; ++p)                                \
            _Pragma("unroll") for (int t = 0; t < RT; ++t) {                              \
                const AV* src = PANEL ? reinterpret_cast<const AV*>(ap + ((int64_t)(s0 + (R)) * NL + p) * lda + (int64_t)(16 * t) * PWC) \
                                      : reinterpret_cast<const AV*>(ap + (int64_t)(16 * t) * lda + kb + p * 4 * E); \
                AA[p][t] = NT ? __builtin_nontemporal_load(src) : *src;                   \
            }                                                                             \
        }
#define LCX_CR_LOADB(R)                                                                   \
        {                                                                                 \
            const f4* src = reinterpret_cast<const f4*>(B + (int64_t)(s0 + (R)) * CHUNK); \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) bst[p] = src[pc];                         \
            }                                                                             \
        }
#define LCX_CR_STOREB(BUF)                                                                \
        {                                                                                 \
            f4* dstp = reinterpret_cast<f4*>(&Bs[BUF][0]);                                \
            _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                             \
                const int pc = p * NTH + (int)threadIdx.x;                                \
                if (PCS % NTH == 0 || pc < PCS) dstp[pc] = bst[p];                        \
            }                                                                             \
        }
#define LCX_CR_MMA(AA, BUF)                                                               \
        if (active) {                                                                     \
            Pk<T, CT> bb[U];                                                              \
            _Pragma("unroll") for (int p = 0; p < NL; ++p)                                \
            _Pragma("unroll") for (int e = 0; e < E; ++e)                                 \
                bb[p * E + e] = *reinterpret_cast<const Pk<T, CT>*>(&Bs[BUF][(p * 4 * E + q * E + e) * Mp + i * CT]); \
            _Pragma("unroll") for (int p = 0; p < NL; ++p)                                \
            _Pragma("unroll") for (int e = 0; e < E; ++e)                                 \
            _Pragma("unroll") for (int t = 0; t < RT; ++t)                                \
            _Pragma("unroll") for (int u = 0; u < CT; ++u)                                \
                acc[t][u] = MF<T>::mma(AA[p][t][e], bb[p * E + e].v[u], acc[t][u]);       \
        }

        LCX_CR_LOADA(0, a0);
        LCX_CR_LOADB(0);
        int r = 0;
        while (true) {
            LCX_CR_STOREB(0);
            if (r + 1 < cnt) { LCX_CR_LOADA(r + 1, a1); LCX_CR_LOADB(r + 1); }
            __syncthreads();
            LCX_CR_MMA(a0, 0);
            if (++r >= cnt) break;
            LCX_CR_STOREB(1);
            if (r + 1 < cnt) { LCX_CR_LOADA(r + 1, a0); LCX_CR_LOADB(r + 1); }
            __syncthreads();
            LCX_CR_MMA(a1, 1);
            if (++r >= cnt) break;
        }
#undef LCX_CR_LOADA
#undef LCX_CR_LOADB
#undef LCX_CR_STOREB
#undef LCX_CR_MMA

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
                    *reinterpret_cast<Pk<T, CT>*>(dst + (16 * t + MF<T>::row(lane, g)) * Mp + i * CT) = o;
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

// ------------------------------------------------------------------------------------------------
// gemm_tn4: the float64 small-shard contraction on v_mfma_f64_4x4x4 (4 blocks).
//
// Measured on MI355X (tools/mfma_peak.hip): v_mfma_f64_16x16x4 issues every ~104 cycles (47.6 TF/s, 60 % of
// the 78.6 TF/s spec) while v_mfma_f64_4x4x4 issues every ~17 cycles for a quarter of the work (72 TF/s).
// Lane layout (tools/mfma_map.hip):  A[blk][i][k] in lane k*16 + blk*4 + i,  B[blk][k][j] in lane k*16 + blk*4 + j,
// D[blk][i][j] in lane i*16 + blk*4 + j.  With the 4 blocks on 4 adjacent groups of 4 columns of X the A operand is
// exactly the 16x16x4 one (lane l: column l & 15, contraction row l >> 4), so X streams global -> VGPR as before;
// the B operand is 4 factors wide and the same for every block, i.e. replicated over blk.  One A register meets
// Mp/4 B registers (factor groups) and Mp/4 one-double accumulators.
//   * same decomposition as gemm_tn: a wave owns 16*RT columns, the block's KW waves split the contraction,
//     grid.y splits write partial tiles, fixed summation orders;
//   * B (Y or W, L2 resident) is staged per wave: global -> VGPR (16 B per lane, coalesced) -> the wave's own
//     LDS strip, rows padded by 32 B so that the 4 contraction rows of a ds_read_b64 land on distinct banks; the
//     operand reads are LDS broadcasts.  No block barrier in the main loop.
// ------------------------------------------------------------------------------------------------
template <int CT, int RT, int KW, int U, bool NT = false, bool SERIAL = false, bool WT = false>
__device__ __forceinline__ void
tn4_body(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, double* __restrict__ out,
         int64_t out_rows, int kgroups /* K / 16 */, int nsplit, const int tile_x, const int split_y) {
    constexpr int Mp = 16 * CT, NG = Mp / 4;
    constexpr int ROWS = 4 * U;                      // rows of B per group
    constexpr int LDB = Mp + 4;                      // padded LDS row (doubles)
    constexpr int PPR = Mp / 2;                      // 16-byte pieces per row
    constexpr int PCS = ROWS * PPR;
    constexpr int PPT = (PCS + 63) / 64;             // pieces per lane
    constexpr int STRIP = 2 * ROWS * LDB;            // doubles per wave (double buffered)
    typedef double d2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4, jj = lane & 3;
    const int64_t v0 = (int64_t)tile_x * (16 * RT);
    const int part = split_y * KW + wave, nparts = nsplit * KW;
    const int ng = kgroups * 4 / U;                  // groups of 4*U rows
    const int g0 = (int)((int64_t)ng * part / nparts), g1 = (int)((int64_t)ng * (part + 1) / nparts);
    const int cnt = g1 - g0;

    double acc[RT][NG];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[t][g] = 0.0;

    const double* ap = A + v0 + (int64_t)kq * lda;
    double* bw = smem + wave * STRIP;
    double a0[U][RT], a1[U][RT];
    d2 bst[PPT];

#define LCX_T4_LOADA(R, AA)                                                                \
    {                                                                                      \
        const int64_t rb = (int64_t)(g0 + (R)) * ROWS;                                     \
        _Pragma("unroll") for (int st = 0; st < U; ++st)                                   \
            load_row_pieces<double, RT, NT>(ap + (rb + 4 * st) * lda, r16, AA[st]);        \
    }
#define LCX_T4_LOADB(R)                                                                    \
    {                                                                                      \
        const d2* src = reinterpret_cast<const d2*>(B + (int64_t)(g0 + (R)) * ROWS * Mp);  \
        _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                                  \
            const int pc = p * 64 + lane;                                                  \
            if (PCS % 64 == 0 || pc < PCS) bst[p] = src[pc];                               \
        }                                                                                  \
    }
#define LCX_T4_STOREB(BUF)                                                                 \
    {                                                                                      \
        _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                                  \
            const int pc = p * 64 + lane;                                                  \
            if (PCS % 64 == 0 || pc < PCS)                                                 \
                *reinterpret_cast<d2*>(bw + (BUF) * ROWS * LDB + (pc / PPR) * LDB + (pc % PPR) * 2) = bst[p]; \
        }                                                                                  \
    }
#define LCX_T4_MMA(AA, BUF)                                                                \
    {                                                                                      \
        _Pragma("unroll") for (int st = 0; st < U; ++st) {                                 \
            const double* brow = bw + (BUF) * ROWS * LDB + (4 * st + kq) * LDB + jj;       \
            double bb[NG];                                                                 \
            _Pragma("unroll") for (int g = 0; g < NG; ++g) bb[g] = brow[4 * g];            \
            _Pragma("unroll") for (int t = 0; t < RT; ++t)                                 \
            _Pragma("unroll") for (int g = 0; g < NG; ++g)                                 \
                acc[t][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(AA[st][t], bb[g], acc[t][g], 0, 0, 0); \
        }                                                                                  \
    }

    if (cnt > 0) {
        LCX_T4_LOADA(0, a0);
        LCX_T4_LOADB(0);
        int r = 0;
        while (true) {
            LCX_T4_STOREB(0);
            if (r + 1 < cnt) { LCX_T4_LOADA(r + 1, a1); LCX_T4_LOADB(r + 1); }
            LCX_T4_MMA(a0, 0);
            if (++r >= cnt) break;
            LCX_T4_STOREB(1);
            if (r + 1 < cnt) { LCX_T4_LOADA(r + 1, a0); LCX_T4_LOADB(r + 1); }
            LCX_T4_MMA(a1, 1);
            if (++r >= cnt) break;
        }
    }
#undef LCX_T4_LOADA
#undef LCX_T4_LOADB
#undef LCX_T4_STOREB
#undef LCX_T4_MMA

    // ---- reduce the KW partial tiles through LDS in a fixed order and write the tile ----------------
    constexpr int TILE = 16 * RT * Mp;
    __syncthreads();                                  // the B strips are dead: the same LDS holds the tiles now
    const int row = ((lane & 15) >> 2) * 4 + (lane >> 4);          // blk*4 + i: the column of X inside the 16-wide piece
    double* dst = out + ((int64_t)split_y * out_rows + v0) * Mp;
    if (SERIAL) {
        // one tile of LDS: the waves add their tiles one after the other (fixed order), so that wide tiles do not
        // cost KW times their size in LDS (occupancy)
        for (int w = 0; w < KW; ++w) {
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        double* p = &smem[piece_col<double, RT>(t, row) * Mp + 4 * g + jj];
                        *p = (w == 0) ? acc[t][g] : *p + acc[t][g];
                    }
            }
            __syncthreads();
        }
        for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) dst[idx] = smem[idx];
        return;
    }
    double* mine = smem + wave * TILE;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            mine[piece_col<double, RT>(t, row) * Mp + 4 * g + jj] = acc[t][g];
    __syncthreads();
    for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) {
        double sacc = smem[idx];
#pragma unroll
        for (int w = 1; w < KW; ++w) sacc += smem[w * TILE + idx];
        if constexpr (WT) __hip_atomic_store(&dst[idx], sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dst[idx] = sacc;
    }
}
template <int CT, int RT, int KW, int U, bool NT = false, bool SERIAL = false>
__global__ void __launch_bounds__(64 * KW)
gemm_tn4_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, double* __restrict__ out,
                int64_t out_rows, int kgroups /* K / 16 */, int nsplit, const int* __restrict__ skip_flag) {
    if (skip_flag != nullptr && *skip_flag != 0) return;
    tn4_body<CT, RT, KW, U, NT, SERIAL>(A, lda, B, out, out_rows, kgroups, nsplit, blockIdx.x, blockIdx.y);
}
// the same pass with in-launch chunk signalling (see ChunkSig): 1-D grid over (row tile, slot), slot fastest
template <int CT, int RT, int KW, int U>
__global__ void __launch_bounds__(64 * KW)
gemm_tn4_sig_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, double* __restrict__ out,
                    int64_t out_rows, int kgroups, int nsplit, ChunkSig sg) {
    const int tile = (int)(blockIdx.x / (unsigned)nsplit), slot = (int)(blockIdx.x % (unsigned)nsplit);
    tn4_body<CT, RT, KW, U, true, false, true>(A, lda, B, out, out_rows, kgroups, nsplit, tile, slot);
    chunk_signal_tail(sg, tile, nsplit);
}
// dynamic LDS of gemm_tn4: max(B strips, reduction tiles)
template <int CT, int RT, int KW, int U, bool SERIAL = false> struct Tn4Lds {
    static constexpr size_t strips = (size_t)KW * 2 * 4 * U * (16 * CT + 4) * sizeof(double);
    static constexpr size_t tiles = (size_t)(SERIAL ? 1 : KW) * 16 * RT * 16 * CT * sizeof(double);
    static constexpr size_t bytes = strips > tiles ? strips : tiles;
};

// ------------------------------------------------------------------------------------------------
// gemm_wide: every contraction of a model with MORE than 256 factors (n_hidden 257..1024; reference linearcorex.py:72 takes any).
//
// The tuned kernels above keep a whole row of m_pad accumulators in registers, which ends at 256 columns.  Beyond that the
// factor axis is tiled like any other: a plain LDS-staged MFMA GEMM with run-time sizes and strides, one kernel for all the
// contractions of the path - untuned by intent (no BASELINE config has more than 128 factors), correct and deterministic.
//   C[z][m][n] = sum_{k in split z} opA(m, k) * B[k][n] (* rowscale[k]),   opA = A[m][k] (TRANS_A false) or A[k][m] (true)
// Block = 4 waves, tile 64 (m) x 64 (n): wave w owns rows 16 w .. 16 w + 15 and four 16-column MFMA tiles; the contraction
// advances 16 elements per step through LDS.  M, N multiples of 64, K multiple of 16 per split; grid = (N / 64, M / 64, splits).
// ------------------------------------------------------------------------------------------------
template <typename T, bool TRANS_A, bool SCALE>
__global__ void __launch_bounds__(256)
gemm_wide_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, const T* __restrict__ rowscale,
                 T* __restrict__ C, int64_t ldc, int64_t M, int64_t K, int nsplit, const int* __restrict__ skip_flag) {
    typedef typename MF<T>::acc_t acc_t;
    __shared__ T As[64][17];          // [m][k], padded: the 16 rows x 4 lane groups of an operand read hit distinct banks
    __shared__ T Bs[16][68];          // [k][n]
    if (skip_flag != nullptr && *skip_flag != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    const int64_t ksteps = K / 16;
    const int64_t s0 = ksteps * blockIdx.z / nsplit, s1 = ksteps * (blockIdx.z + 1) / nsplit;
    acc_t acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = (acc_t){0, 0, 0, 0};
    // staging: thread -> one 4-element vector of A (TRANS_A: row k = tid / 16, 4 consecutive m; else row m = tid / 4, 4 consecutive k)
    // and one of B (row k = tid / 16, 4 consecutive n); the vectors of step st + 1 are in flight while step st multiplies
    const int ak = TRANS_A ? (tid >> 4) : (tid & 3) * 4, am = TRANS_A ? (tid & 15) * 4 : (tid >> 2);
    const int bk = tid >> 4, bn = (tid & 15) * 4;
    const T* asrc = TRANS_A ? A + (int64_t)ak * lda + m0 + am : A + (m0 + am) * lda + ak;
    const T* bsrc = B + (int64_t)bk * ldb + n0 + bn;
    Pk<T, 4> a4, b4;
    if (s0 < s1) {
        a4 = ldg<T, 4>(TRANS_A ? asrc + s0 * 16 * lda : asrc + s0 * 16);
        b4 = ldg<T, 4>(bsrc + s0 * 16 * ldb);
    }
    for (int64_t st = s0; st < s1; ++st) {
        const int64_t k0 = st * 16;
        __syncthreads();
        if (TRANS_A) {
            const T sc = SCALE ? rowscale[k0 + ak] : (T)1;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[am + e][ak] = a4.v[e] * sc;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) As[am][ak + e] = a4.v[e] * (SCALE ? rowscale[k0 + ak + e] : (T)1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[bk][bn + e] = b4.v[e];
        __syncthreads();
        if (st + 1 < s1) {
            a4 = ldg<T, 4>(TRANS_A ? asrc + (k0 + 16) * lda : asrc + k0 + 16);
            b4 = ldg<T, 4>(bsrc + (k0 + 16) * ldb);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const T av = As[16 * wave + i][4 * s + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = MF<T>::mma(av, Bs[4 * s + q][16 * u + i], acc[u]);
        }
    }
    T* dst = C + ((int64_t)blockIdx.z * M + m0 + 16 * wave) * ldc + n0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[(int64_t)MF<T>::row(lane, g) * ldc + 16 * u + i] = acc[u][g];
}

// tile shapes of gemm_ct: 16-byte A loads wherever the accumulators fit (RT*CT*4 registers of T)
template <typename T, int CT> struct CtShape {
    // 256 padded factors (CT = 16): half the column tile, so that the accumulators stay at 128 registers
    static constexpr int RT = (sizeof(T) == 8) ? (CT >= 16 ? 1 : (CT >= 4 ? 2 : 4)) : (CT >= 16 ? 2 : 4);
    // measured: KW=4, U=4 is the shape at 64 columns (tools/gemm_probe4, 50k x 20k float32: 122 TF/s; 8 waves lose 6 % there, U=2
    // loses more).  At 128 float32 columns 8 waves share one copy of B per block instead of 2 x 4 waves on two (B is 8 KB per group
    // against 4 KB of X per wave, so halving its re-reads shows): +2-3 % on the panel-major copy (profiles/r04_gemm_probe4_panelsweep.txt:
    // X.B^T 11 064 vs 11 263 us, X^T.Y 11 040 vs 11 271 us at the config-4 shard) and on the row-major layouts alike
    // (profiles/r05_layout_ab_one_box.txt: 20.1-20.2 vs 19.6-19.7 it/s)
    static constexpr int KW = (sizeof(T) == 4 && CT == 8) ? 8 : 4;
    static constexpr int U = 4;
};

// tile shapes per (dtype, CT): chosen so accumulators + two register sets stay under ~200 VGPRs
template <typename T, int CT> struct NtShape { static constexpr int RT = (sizeof(T) == 8 && CT >= 8) ? 1 : 2; };
// tn: wave tile = 16*RT columns of A.  Measured on MI355X (tools/gemm_probe, 10k x 5k f64, Mp=32):
// 64-column tiles (512 B per row per wave) stream ~8 % faster than 32-column ones.
template <typename T, int CT> struct TnShape {
    static constexpr int RT = (sizeof(T) == 8) ? (CT >= 8 ? 1 : (CT >= 4 ? 2 : 4)) : (CT >= 8 ? 2 : 4);
};

}  // namespace lcx
