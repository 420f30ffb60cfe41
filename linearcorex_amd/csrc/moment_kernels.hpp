// moment_kernels.hpp - per-variable / per-factor kernels of the Linear CorEx fit loop (gfx950).
//
// Every "m by nv" array of the reference (W, rho, rhoinvrho, Qij, grad, update ...) is stored
// variable-major on the device: [Vp][Mp], Mp = n_hidden padded to a multiple of 16, so that
//   * a variable's Mp-vector is contiguous (coalesced per-variable math, natural nv-sharding),
//   * the array is directly the "B" operand of gemm_nt and the output layout of gemm_tn.
// Padded factors (j >= m) carry W = 0 and stay exactly 0 through every formula; padded variables
// (v >= V) are never written and stay 0.
//
// These kernels are HBM-bound on M x V arrays (SURVEY.md 8a "K3-K5"): one thread per (variable,
// factor), Mp consecutive threads per variable, wavefront shuffle reductions over the factor axis,
// the m x m operators (ry, H) staged in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_kernels.hpp"       // tn_body (gram_tc_kernel)

namespace lcx {

constexpr int PV_THREADS = 256;
// The m x m operators of the per-variable kernels (ry, H, ry^-1) are staged in LDS up to 128 padded factors; above
// (n_hidden 129..256) Mp^2 elements no longer fit next to the per-variable strips and every thread reads its column of
// the operator from global memory instead (coalesced over the factor index, served by L2).
template <int Mp> struct OpInLds { static constexpr bool v = Mp <= 128; };
// Threads per block of the per-variable kernels: 256 (PV_THREADS / Mp variables per block) up to 256 padded factors; beyond
// (n_hidden 257..1024, the untuned wide path) one variable per block and one thread per factor.
template <int Mp> struct Pvt { static constexpr int v = Mp > 256 ? Mp : 256; };

// state scalars per moment set (mirrors LCX_S_* in include/lcx.h)
struct SetState {
    double tc, max_uj, invalid_d, tangent, sum_log_rj, r5, r6, r7;
    int invalid;          // read by the GEMM kernels as skip flag
    unsigned int seq;     // publication counter of the pinned host mirror
    int pad[14];
};

// Copy a set's scalars to its pinned host mirror and bump the mirror's sequence number last, so
// that the host can poll the mirror instead of issuing a device-to-host copy + stream sync.
__device__ __forceinline__ void publish_state(const SetState* st, SetState* host, unsigned int seq) {
    // The mirror lives in fine-grained pinned host memory: write-through system-scope stores, drained, then the
    // sequence number (posted writes of one device to one host page stay ordered).  No release fence here: when
    // this runs in the last block of a kernel that has just written megabytes, a system-scope release first
    // writes back every dirty line of the XCD's L2 (measured: +7-8 us on moments_epilogue / update_kernel).
    __hip_atomic_store(&host->tc, st->tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host->max_uj, st->max_uj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host->invalid_d, st->invalid_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host->tangent, st->tangent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host->sum_log_rj, st->sum_log_rj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host->invalid, st->invalid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&host->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// "Am I the last block?" for a grid of n blocks, called by ONE thread per block after its partial results are
// stored (write-through atomics) and drained.  A single counter serialises every arrival (~12 ns each: 7-9 us for
// 600-800 blocks, measured); blocks are dispatched round-robin over the 8 XCDs, so 8 counters keyed by b & 7 take
// the arrivals in parallel and only the last arriver of each one touches the top counter.  tk: 9 zeroed words, left
// zeroed.  Correct for any placement; the b & 7 key is only for speed.
__device__ __forceinline__ bool arrive_last(unsigned int* tk, unsigned int b, unsigned int n) {
    const unsigned int x = b & 7u;
    const unsigned int nx = (n - x + 7u) >> 3;                     // blocks with index = x (mod 8)
    const unsigned int t = __hip_atomic_fetch_add(&tk[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != nx - 1u) return false;
    __hip_atomic_store(&tk[x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int groups = n < 8u ? n : 8u;
    const unsigned int t2 = __hip_atomic_fetch_add(&tk[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t2 != groups - 1u) return false;
    __hip_atomic_store(&tk[8], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// replicated per-factor quantities of a moment set
struct SmallDesc {
    double* uj;    // [Mp]
    double* ry;    // [Mp*Mp], diagonal forced to 1 (linearcorex.py:263)
    double* wmag;  // [Mp]  sum_i W_ji^2 (linearcorex.py:130, :249)
};

// sum over the Mp consecutive threads that share a variable. Mp in {16,32,64,128}.
template <int Mp, typename R>
__device__ __forceinline__ R group_sum(R v, R* scratch /* [PV_THREADS/64] per use */, int tid) {
    constexpr int W = Mp < 64 ? Mp : 64;
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, W);
    if (Mp > 64) {
        // Mp / 64 waves per variable: combine through LDS in a fixed order
        constexpr int NW = Mp / 64;
        __syncthreads();
        if ((tid & 63) == 0) scratch[tid >> 6] = v;
        __syncthreads();
        const int base = ((tid >> 6) / NW) * NW;
        v = scratch[base];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += scratch[base + w];
    }
    return v;
}

template <typename R, int NT = PV_THREADS>
__device__ __forceinline__ R block_sum(R v, R* scratch /* [NT/64] */, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    R s = scratch[0];
    for (int w = 1; w < NT / 64; ++w) s += scratch[w];
    return s;
}

// ------------------------------------------------------------------------------------------------
// sum grid-level partials: out[i] = sum_s in[s][i]   (OUT may be double for the exchange buffer)
// ------------------------------------------------------------------------------------------------
template <typename T, typename OUT>
__global__ void reduce_partials_kernel(const T* __restrict__ in, int nsplit, int64_t n, int64_t stride,
                                       OUT* __restrict__ out, const int* __restrict__ skip_flag,
                                       OUT* __restrict__ out2 = nullptr) {
    if (skip_flag != nullptr && *skip_flag != 0) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T s = in[i];
        for (int k = 1; k < nsplit; ++k) s += in[k * stride + i];
        out[i] = (OUT)s;
        if (out2 != nullptr) out2[i] = (OUT)s;       // the set's own copy of Y (no separate copy launch)
    }
}

// ------------------------------------------------------------------------------------------------
// per-factor moments from the Gram matrices (one block):
//   ry = (1-eps^2) Y^T Y / N + eps^2 W W^T   (== ws.dot(rho.T), linearcorex.py:261)
//   uj = diag(ry) before the diagonal is set to 1 (:249, :263); early-exit flag (:250-251)
// gy: [nsplit][Mp][Mp] partials of Y^T Y; gw: [Mp][Mp] (already summed over shards)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
small_moments_kernel(const T* __restrict__ gy, int nsplit, const T* __restrict__ gw, int nsplit_w,
                     int Mp, int m, double n_samples, double eps, int quick, SmallDesc sm,
                     SetState* st, unsigned int* ticket) {
    // Each block sums the partial Gram tiles of 32 matrix elements (8 threads per element, fixed
    // order) and writes ry / uj; the block that draws the last ticket then derives the per-factor
    // scalars.  One launch instead of reduce + finish; deterministic (no floating-point atomics).
    __shared__ T shy[8][32];
    __shared__ T shw[8][32];
    __shared__ double lg[256];
    __shared__ double uu[256];
    __shared__ int last_s;
    const int tid = threadIdx.x, e = tid & 31, g = tid >> 5;
    const int64_t mm = (int64_t)Mp * Mp;
    const int64_t idx = (int64_t)blockIdx.x * 32 + e;
    T a = (T)0, b = (T)0;
    for (int k = g; k < nsplit; k += 8) a += gy[k * mm + idx];
    for (int k = g; k < nsplit_w; k += 8) b += gw[k * mm + idx];
    shy[g][e] = a;
    shw[g][e] = b;
    __syncthreads();
    if (g == 0) {
        T gyv = shy[0][e], gwv = shw[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) { gyv += shy[k][e]; gwv += shw[k][e]; }
        const T c1 = (T)(1.0 - eps * eps), c2 = (T)(eps * eps), ns = (T)n_samples;
        const T val = c1 * gyv / ns + c2 * gwv;
        const int j = (int)(idx / Mp), k2 = (int)(idx % Mp);
        if (j == k2) {
            sm.uj[j] = (double)val;
            sm.wmag[j] = (double)gwv;
            sm.ry[idx] = 1.0;
        } else {
            sm.ry[idx] = (double)val;
        }
    }
    // hand-off to the last block: release (agent scope) -> ticket; last block acquires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    {
        // thread t takes the factors t, t + 256, ... (one each up to 256 factors): max uj and sum of log(1 - uj)
        double u = -1e300, l = 0.0;
        for (int j = tid; j < m; j += 256) {
            const double uv = __hip_atomic_load(&sm.uj[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u = fmax(u, uv);
            l += (double)log((T)1 - (T)uv);        // log(1-uj) in working precision (:274)
        }
        uu[tid] = u;
        lg[tid] = l;
    }
    __syncthreads();
    if (tid == 0) {
        double mx = -1e300, slog = 0.0;
        const int nact = m < 256 ? m : 256;
        for (int j = 0; j < nact; ++j) { mx = fmax(mx, uu[j]); slog += lg[j]; }
        st->max_uj = mx;
        st->sum_log_rj = slog;
        const int inv = (quick && mx >= 1.0) ? 1 : 0;
        st->invalid = inv;
        st->invalid_d = (double)inv;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// TC = sum log(1+Si) - 1/2 sum log(1+QiSi2) + 1/2 sum log(1-uj), rounded to the working precision
template <typename T>
__device__ __forceinline__ void tc_store(const double* sbuf, SetState* st) {
    if (st->invalid) { st->tc = __builtin_nan(""); return; }
    const T tc = (T)sbuf[0] - (T)0.5 * (T)sbuf[1] + (T)0.5 * (T)st->sum_log_rj;
    st->tc = (double)tc;
}
// ------------------------------------------------------------------------------------------------
// moments epilogue: from D = X^T Y partials to rho, rhoinvrho, Qij, Si, Qi-Si^2 and the two log sums
// (linearcorex.py:260, :264-269, :272-273).  grid-stride over variable groups.
// dynamic LDS: ry_s[Mp*Mp] (T) + rir_s[VPB*Mp] (T)
// ------------------------------------------------------------------------------------------------
// The two log sums leave the kernel as per-block pairs in tcpart (plain stores): they are summed, TC is formed and the
// state is published by the tail block that rides in the launch of the H Gram that always follows (gram_tc_kernel).
// A fused tail here (ticket, last block sums and publishes) cost 11-14 us on top of an 8-16 us body: two dependent
// atomic round trips plus the publication (tools/epilogue_probe.hip, profiles/r01_epilogue_probe_tail.txt).
template <typename T, int Mp>
__global__ void __launch_bounds__(Pvt<Mp>::v)
moments_epilogue_kernel(const T* __restrict__ dpart, int nsplit, int64_t pstride,
                        const T* __restrict__ d_base, const T* __restrict__ d_dir, T eta,
                        T* __restrict__ d_out,
                        const T* __restrict__ W, const double* __restrict__ ry, int64_t V,
                        double n_samples, double eps, T* __restrict__ rho_o, T* __restrict__ rir_o,
                        T* __restrict__ qij_o, T* __restrict__ si_o, T* __restrict__ q2_o,
                        T* __restrict__ hscale_o, double* __restrict__ tcpart,
                        const int* __restrict__ skip_flag) {
    constexpr int NT = Pvt<Mp>::v;
    constexpr int VPB = NT / Mp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* ry_s = reinterpret_cast<T*>(smem_raw);
    T* rir_s = ry_s + (OpInLds<Mp>::v ? Mp * Mp : 0);
    __shared__ T gs_scratch[NT / 64];
    __shared__ double bs_scratch[NT / 64];
    if (skip_flag != nullptr && *skip_flag != 0) return;       // invalid trial (:250-251): the tail block still publishes

    const int tid = threadIdx.x, vl = tid / Mp, j = tid % Mp;
    if (OpInLds<Mp>::v)
        for (int idx = tid; idx < Mp * Mp; idx += NT) ry_s[idx] = (T)ry[idx];
    __syncthreads();

    const T c1 = (T)(1.0 - eps * eps), c2 = (T)(eps * eps), ns = (T)n_samples;
    double s1 = 0.0, s2 = 0.0;
    const int64_t ngroups = (V + VPB - 1) / VPB;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t v = grp * VPB + vl;
        const bool ok = v < V;
        const int64_t o = (ok ? v : 0) * Mp + j;
        // D = X^T.Y of this weight matrix: either the partial sums of a fresh pass over X, or - X^T.Y
        // being linear in W - D(W) + eta*D(update) from the current solution
        T d;
        if (d_base != nullptr) {
            d = d_base[o] + eta * d_dir[o];
        } else {
            d = dpart[o];
            for (int k = 1; k < nsplit; ++k) d += dpart[k * pstride + o];
        }
        if (ok) d_out[o] = d;
        const T rho = ok ? (c1 * d / ns + c2 * W[o]) : (T)0;
        const T inv = (T)1 / ((T)1 - rho * rho);
        const T rir = rho * inv;
        __syncthreads();                       // rir_s reuse across iterations
        rir_s[vl * Mp + j] = rir;
        const T si = group_sum<Mp, T>(rho * rir, gs_scratch, tid);
        __syncthreads();
        T qv = (T)0;
        if (OpInLds<Mp>::v) {
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) qv += ry_s[k * Mp + j] * rir_s[vl * Mp + k];   // ry symmetric
        } else {
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) qv += (T)ry[k * Mp + j] * rir_s[vl * Mp + k];
        }
        const T q2 = group_sum<Mp, T>(rir * (qv - si * rho), gs_scratch, tid);
        if (ok) {
            rho_o[o] = rho;
            rir_o[o] = rir;
            qij_o[o] = qv;
            if (j == 0) {
                si_o[v] = si;
                q2_o[v] = q2;
                hscale_o[v] = (T)1 / ((T)1 + q2);
                s1 += (double)log((T)1 + si);
                s2 += (double)log((T)1 + q2);
            }
        }
    }
    s1 = block_sum<double, NT>(s1, bs_scratch, tid);
    s2 = block_sum<double, NT>(s2, bs_scratch, tid);
    if (tid == 0) {
        tcpart[2 * blockIdx.x] = s1;
        tcpart[2 * blockIdx.x + 1] = s2;
    }
}

// out[k] = sum_b part[b][k], one block per output value k, fixed summation order (deterministic)
template <typename OUT>
__global__ void __launch_bounds__(PV_THREADS)
sum_partials_kernel(const double* __restrict__ part, int nblocks, int nvals, OUT* __restrict__ out,
                    const int* __restrict__ skip_flag) {
    __shared__ double bs_scratch[PV_THREADS / 64];
    if (skip_flag != nullptr && *skip_flag != 0) return;
    const int k = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int b = tid; b < nblocks; b += PV_THREADS) s += part[(int64_t)b * nvals + k];
    s = block_sum<double>(s, bs_scratch, tid);
    if (tid == 0) out[k] = (OUT)s;
}

// out[i] = sum_s in[s][i] when there are many splits (few column tiles, e.g. n_samples << n_variables): 8 threads share
// an element and sum every 8th slot, then the 8 sums are added in fixed order.  One thread per element would issue
// nsplit dependent-latency loads (measured: 64 slots of 448 x 32: 16 us; this form ~5 us).
template <typename T, typename OUT>
__device__ __forceinline__ void wide_sum_block(const T* in, int nsplit, int64_t n, int64_t stride, int64_t block,
                                               OUT* out /* may be slot 0 of `in` */, OUT* out2, T (*sh)[32]) {
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = block * 32 + e;
    T s = (T)0;
    if (i < n) {
#pragma unroll 4
        for (int k = g; k < nsplit; k += 8) s += in[k * stride + i];
    }
    sh[g][e] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        T t = sh[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][e];
        out[i] = (OUT)t;
        if (out2 != nullptr) out2[i] = (OUT)t;
    }
}
constexpr int WIDE_SPLITS = 12;      // from this many slots on the reductions of Y use the 8-threads-per-element form

template <typename T, typename OUT>
__global__ void __launch_bounds__(256)
reduce_partials_wide_kernel(const T* in, int nsplit, int64_t n, int64_t stride, OUT* out,
                            const int* __restrict__ skip_flag, OUT* out2) {
    __shared__ T sh[8][32];
    if (skip_flag != nullptr && *skip_flag != 0) return;
    wide_sum_block<T, OUT>(in, nsplit, n, stride, blockIdx.x, out, out2, sh);
}

// One launch for the two reductions that follow the X.grad^T pass: blocks [0, yblocks) sum the
// grid-level partial tiles of Y (skipped when nsplit == 1), the next nvals blocks sum the per-block
// Bj partials into the tail of the exchange buffer.  WIDE: the Y blocks each own 32 elements (8 threads per element).
template <typename T, bool WIDE = false>
__global__ void __launch_bounds__(PV_THREADS)
reduce_y_bj_kernel(const T* __restrict__ ypart, int nsplit, int64_t n, T* __restrict__ yout, int yblocks,
                   const double* __restrict__ part, int nblocks, int nvals, T* __restrict__ tail) {
    __shared__ double bs_scratch[PV_THREADS / 64];
    __shared__ T sh[WIDE ? 8 : 1][32];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < yblocks) {
        if (WIDE) {
            wide_sum_block<T, T>(ypart, nsplit, n, n, blockIdx.x, yout, (T*)nullptr, sh);
            return;
        }
        for (int64_t i = (int64_t)blockIdx.x * PV_THREADS + tid; i < n; i += (int64_t)yblocks * PV_THREADS) {
            T s = ypart[i];
            for (int k = 1; k < nsplit; ++k) s += ypart[k * n + i];
            yout[i] = s;
        }
        return;
    }
    const int k = blockIdx.x - yblocks;
    double s = 0.0;
    for (int b = tid; b < nblocks; b += PV_THREADS) s += part[(int64_t)b * nvals + k];
    s = block_sum<double>(s, bs_scratch, tid);
    if (tid == 0) tail[k] = (T)s;
}

// out[i] = sum_s in[s][i] for many splits: 8 threads share an element, fixed order
template <typename T, typename OUT>
__global__ void __launch_bounds__(256)
reduce_wide_kernel(const T* __restrict__ in, int nsplit, int64_t n, int64_t stride,
                   OUT* __restrict__ out, const int* __restrict__ skip_flag) {
    __shared__ T sh[8][32];
    if (skip_flag != nullptr && *skip_flag != 0) return;
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + e;
    T s = (T)0;
    if (i < n)
        for (int k = g; k < nsplit; k += 8) s += in[k * stride + i];
    sh[g][e] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        T t = sh[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][e];
        out[i] = (OUT)t;
    }
}

template <typename T>
__global__ void tc_final_kernel(const double* __restrict__ sbuf, SetState* st, SetState* host, unsigned int seq) {
    tc_store<T>(sbuf, st);
    st->tangent = sbuf[2];       // update_tangent (:305) of the direction this trial belongs to, now global
    publish_state(st, host, seq);
}
// ------------------------------------------------------------------------------------------------
// gradient (linearcorex.py:293-300) and the per-block partial of Bj (:302).
// dynamic LDS: h_s[Mp*(Mp+1)] (T) + w_s[VPB*Mp] (T) + bj_s[VPB*Mp] (double)
// ------------------------------------------------------------------------------------------------
template <typename T, int Mp>
__global__ void __launch_bounds__(Pvt<Mp>::v)
grad_kernel(const T* __restrict__ W, const T* __restrict__ rho_i, const T* __restrict__ rir_i,
            const T* __restrict__ qij_i, const T* __restrict__ si_i, const T* __restrict__ q2_i,
            const double* __restrict__ uj, const double* __restrict__ H /* sbuf, diag ignored */,
            int64_t V, T* __restrict__ grad_o, double* __restrict__ bjpart, T* __restrict__ grad2_o = nullptr) {
    constexpr int NT = Pvt<Mp>::v;
    constexpr int VPB = NT / Mp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* h_s = reinterpret_cast<T*>(smem_raw);        // [Mp][Mp+1]
    T* w_s = h_s + (OpInLds<Mp>::v ? Mp * (Mp + 1) : 0);           // [VPB][Mp]
    double* bj_s = reinterpret_cast<double*>(w_s + VPB * Mp + (((VPB * Mp) & 1) ? 1 : 0));
    const int tid = threadIdx.x, vl = tid / Mp, j = tid % Mp;
    if (OpInLds<Mp>::v)
        for (int idx = tid; idx < Mp * Mp; idx += NT) {
            const int a = idx / Mp, b = idx % Mp;
            h_s[a * (Mp + 1) + b] = (a == b) ? (T)0 : (T)H[idx];       // fill_diagonal(H, 0), :295
        }
    const T rj = (T)1 - (T)uj[j];
    __syncthreads();
    double bj = 0.0;
    const int64_t ngroups = (V + VPB - 1) / VPB;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t v = grp * VPB + vl;
        const bool ok = v < V;
        const int64_t o = (ok ? v : 0) * Mp + j;
        const T w = ok ? W[o] : (T)0;
        __syncthreads();
        w_s[vl * Mp + j] = w;
        __syncthreads();
        const T rho = rho_i[o], rir = rir_i[o], qij = qij_i[o];
        const T si = si_i[ok ? v : 0], q2 = q2_i[ok ? v : 0];
        const T inv = (T)1 / ((T)1 - rho * rho);
        T g = w / rj;                                                           // :296
        g -= (T)2 * inv * rir / ((T)1 + si);                                    // :297
        g += inv * inv * (((T)1 + rho * rho) * qij - (T)2 * rho * si) / ((T)1 + q2);   // :298-299
        T hw = (T)0;
        if (OpInLds<Mp>::v) {
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) hw += h_s[j * (Mp + 1) + k] * w_s[vl * Mp + k];
        } else {
            // H is a Gram matrix (symmetric up to rounding): column j is read, coalesced over j
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) hw += (k == j ? (T)0 : (T)H[k * Mp + j]) * w_s[vl * Mp + k];
        }
        g += hw;                                                                // :300
        if (ok) {
            grad_o[o] = g;
            if (grad2_o != nullptr) grad2_o[v * (2 * Mp) + j] = g;          // columns [0, Mp) of the merged operand [V][2 Mp]
            bj += (double)(rho * g);
        }
    }
    bj_s[vl * Mp + j] = bj;
    __syncthreads();
    if (tid < Mp) {
        double s = bj_s[tid];
        for (int k = 1; k < VPB; ++k) s += bj_s[k * Mp + tid];
        bjpart[(int64_t)blockIdx.x * Mp + tid] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// second half of _sig (:212), update (:303) and the per-block partial of update_tangent (:305)
// ------------------------------------------------------------------------------------------------
template <typename T, int Mp>
__global__ void __launch_bounds__(PV_THREADS)
update_kernel(const T* __restrict__ dpart, int nsplit, int64_t pstride, const T* __restrict__ grad,
              const T* __restrict__ W, const double* __restrict__ uj, const T* __restrict__ bj_tail,
              int64_t V, double n_samples, double eps, T* __restrict__ update_o,
              T* __restrict__ sgrad_o, double* __restrict__ tanpart,
              const T* __restrict__ d_cur, T* __restrict__ d_dir_o,
              int update_blocks, const T* __restrict__ yg, const T* __restrict__ ycur, int64_t ny,
              T* __restrict__ ydir_o, T* __restrict__ w1_o, int n_ranks, T* __restrict__ w1b_o = nullptr,
              int tan_offset = 0) {
    __shared__ double bs_scratch[PV_THREADS / 64];
    const int tid = threadIdx.x;
    const T c1 = (T)(1.0 - eps * eps), c2 = (T)(eps * eps), ns = (T)n_samples;
    // nsplit == 0: the X^T.Y_g pass was not run.  update_tangent (:305) = sum_ji sig_grad_ji update_ji with
    // sig_grad = (1-eps^2) X^T(X grad^T)^T / N + eps^2 grad (:212) equals
    //     (1-eps^2)/N <Y_g, Y(update)>  +  eps^2 <grad, update>,        Y(update) = X.update^T = -rj (Y_g - c Y),
    // and every term of that is in hand after the FIRST pass of _sig: the second pass only ever fed this scalar.
    const bool yspace = nsplit == 0;
    double tan = 0.0;
    if ((int)blockIdx.x >= update_blocks) {
        // the blocks past `update_blocks` form Y(update) = -rj (Y_g - c Y) on [Npad][Mp]
        const int nb = gridDim.x - update_blocks;
        for (int64_t i = (int64_t)(blockIdx.x - update_blocks) * PV_THREADS + tid; i < ny; i += (int64_t)nb * PV_THREADS) {
            const int j = (int)(i % Mp);
            const T rj = (T)1 - (T)uj[j];
            const T ygi = yg[i];
            const T yd = -rj * (ygi - (T)2 * bj_tail[j] / ((T)2 - rj) * ycur[i]);
            ydir_o[i] = yd;
            if (yspace) tan += (double)(ygi * yd);
        }
        // Y_g and Y are replicated on every rank while the partials are summed over ranks
        tan *= (double)c1 / n_samples / (double)n_ranks;
    } else {
        const int64_t total = V * Mp;
        for (int64_t o = (int64_t)blockIdx.x * PV_THREADS + tid; o < total;
             o += (int64_t)update_blocks * PV_THREADS) {
            const int j = (int)(o % Mp);
            const T g = grad[o];
            const T rj = (T)1 - (T)uj[j];
            const T up = -rj * (g - (T)2 * W[o] / ((T)2 - rj) * bj_tail[j]);         // :303
            update_o[o] = up;
            if (w1_o != nullptr) w1_o[o] = W[o] + up;                               // :320 at eta = 1 (the first trial)
            if (w1b_o != nullptr) w1b_o[(o / Mp) * (2 * Mp) + Mp + j] = W[o] + up;   // columns [Mp, 2 Mp) of the merged operand
            if (yspace) {
                tan += (double)(c2 * g * up);
            } else {
                T d = dpart[o];
                for (int k = 1; k < nsplit; ++k) d += dpart[k * pstride + o];
                const T sg = c1 * d / ns + c2 * g;                                  // :212
                sgrad_o[o] = sg;
                // update_j = -rj (grad_j - c_j W_j), c_j = 2 Bj / (2 - rj), is a per-factor combination of
                // grad and W, and X^T.(X.u^T) acts row-wise and linearly, so D(update) needs no pass over X
                if (d_dir_o != nullptr) d_dir_o[o] = -rj * (d - (T)2 * bj_tail[j] / ((T)2 - rj) * d_cur[o]);
                tan += (double)(sg * up);
            }
        }
    }
    // per-block partial of update_tangent: summed by the tail block of the first trial's evaluation (gram_tc_kernel),
    // or by tangent_finalize_kernel when somebody asks for the state before that
    tan = block_sum<double>(tan, bs_scratch, tid);
    if (tid == 0) tanpart[tan_offset + blockIdx.x] = tan;
}

// Merged pass (one read of X for X.grad^T and X.(ws + update)^T): sum the partial slots of Y2 [slots][Npad][2 Mp] and
// split the columns: [0, Mp) -> Y_g = X.grad^T (:210), [Mp, 2 Mp) -> Y of the first trial (:321), the latter into the
// exchange buffer and into the trial set's own copy.
template <typename T>
__global__ void reduce_split_kernel(const T* __restrict__ in, int nsplit, int64_t n2, int Mp, T* __restrict__ yg,
                                    T* __restrict__ y1, T* __restrict__ y1b) {
    const int w = 2 * Mp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        T s = in[i];
        for (int k = 1; k < nsplit; ++k) s += in[k * n2 + i];
        const int64_t r = i / w;
        const int c = (int)(i - r * w);
        if (c < Mp) yg[r * Mp + c] = s;
        else { y1[r * Mp + c - Mp] = s; y1b[r * Mp + c - Mp] = s; }
    }
}


// ------------------------------------------------------------------------------------------------
// Tail of a moment evaluation, run by ONE block that rides in another launch (gram_tc_kernel): sums the per-block log
// sums of moments_epilogue_kernel into sbuf[0..1], a pending update_tangent (per-block partials of update_kernel)
// into sbuf[2], and - with one GPU - forms TC and publishes the state to the host mirror.  Fixed summation order for a
// given block size: deterministic.
// ------------------------------------------------------------------------------------------------
struct TcTail {
    const double* tcpart; int n_tc;          // (s1, s2) pairs, one per epilogue block
    const double* tanpart; int n_tan;        // tangent partials of the direction in flight (0: none pending)
    double* sbuf;
    SetState* st;                            // state of the evaluated set
    SetState* st_cur;                        // state of the current solution (owner of the tangent)
    SetState* host;
    unsigned int seq;
    int single;                              // one GPU: nothing to exchange, publish here
    const int* skip_flag;                    // invalid trial: no sums were written
};

// sum over the block (any multiple of 64 threads up to 512), result valid in thread 0
__device__ __forceinline__ double tail_block_sum(double v, double* scratch /* [8] */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = scratch[0];
    const int nw = (int)blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) s += scratch[w];
    return s;
}

template <typename T>
__device__ __forceinline__ void tc_tail_block(const TcTail& t) {
    __shared__ double tail_scratch[8];
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool skipped = t.skip_flag != nullptr && *t.skip_flag != 0;
    double a1 = 0.0, a2 = 0.0, tg = 0.0;
    if (!skipped)
        for (int b = tid; b < t.n_tc; b += nt) { a1 += t.tcpart[2 * b]; a2 += t.tcpart[2 * b + 1]; }
    for (int b = tid; b < t.n_tan; b += nt) tg += t.tanpart[b];
    a1 = tail_block_sum(a1, tail_scratch);
    a2 = tail_block_sum(a2, tail_scratch);
    if (t.n_tan > 0) tg = tail_block_sum(tg, tail_scratch);
    if (tid != 0) return;
    if (!skipped) { t.sbuf[0] = a1; t.sbuf[1] = a2; }
    if (t.n_tan > 0) {
        t.sbuf[2] = tg;
        if (t.single) t.st_cur->tangent = tg;
    }
    if (t.single) {                 // nothing to exchange: TC, the tangent of the direction in flight, host mirror
        tc_store<T>(t.sbuf, t.st);
        if (t.st != t.st_cur) t.st->tangent = t.st_cur->tangent;
        publish_state(t.st, t.host, t.seq);
    }
}

// H partial Gram (rhoinvrho scaled by 1/(1+Qi-Si^2), linearcorex.py:294) in blockIdx.z == 0, the tail of the moment
// evaluation in the single block z == 1: the host sees TC a few microseconds into this launch instead of at the end
// of a serial ticket / last-block chain in the epilogue.
template <typename T, int CT, int RT, int KW>
__global__ void __launch_bounds__(64 * KW)
gram_tc_kernel(const T* __restrict__ A, const T* __restrict__ rowscale, T* __restrict__ out, int kgroups, int nsplit,
               const int* __restrict__ skip_flag, TcTail tail) {
    if (blockIdx.z == 1) {
        if (blockIdx.x == 0 && blockIdx.y == 0) tc_tail_block<T>(tail);
        return;
    }
    if (skip_flag != nullptr && *skip_flag != 0) return;
    tn_body<T, CT, RT, KW, true, 4>(A, 16 * CT, 16 * RT, A, rowscale, out, 16 * CT, kgroups, nsplit, blockIdx.x, blockIdx.y);
}

// the tail as a launch of its own (the wide path, whose H Gram runs on gemm_wide)
template <typename T>
__global__ void __launch_bounds__(256) tc_tail_kernel(TcTail tail) { tc_tail_block<T>(tail); }

// update_tangent on demand (lcx_read_state of the current solution before any trial was evaluated)
template <typename T>
__global__ void __launch_bounds__(256)
tangent_finalize_kernel(const double* __restrict__ tanpart, int n_tan, double* __restrict__ sbuf, SetState* st, SetState* host,
                        unsigned int seq, int single) {
    __shared__ double tail_scratch[8];
    double tg = 0.0;
    for (int b = threadIdx.x; b < n_tan; b += blockDim.x) tg += tanpart[b];
    tg = tail_block_sum(tg, tail_scratch);
    if (threadIdx.x == 0) {
        sbuf[2] = tg;
        if (single) { st->tangent = tg; publish_state(st, host, seq); }
    }
}

// out = a + eta * b  (Y of a line-search trial from Y and Y(update))
// w_update = ws + eta * update (:320)
template <typename T>
__global__ void axpy_kernel(const T* __restrict__ w, const T* __restrict__ up, T eta, int64_t n,
                            T* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = w[i] + eta * up[i];
}

// two axpys in one launch: w_update = ws + eta*update (:320) and Y' = Y + eta*Y(update)
template <typename T>
__global__ void axpy2_kernel(const T* __restrict__ a1, const T* __restrict__ b1, T* __restrict__ o1, int64_t n1,
                             const T* __restrict__ a2, const T* __restrict__ b2, T* __restrict__ o2, int64_t n2,
                             T eta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n1) o1[i] = a1[i] + eta * b1[i];
        else o2[i - n1] = a2[i - n1] + eta * b2[i - n1];
    }
}

// stage change (:130-133): ws *= 0.001*floor(1000*a_j)
template <typename T>
__global__ void rescale_kernel(T* __restrict__ w, int64_t n, int Mp, const double* __restrict__ uj,
                               const double* __restrict__ wmag, double eps_old, double eps_new) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Mp);
        const T u = (T)uj[j];
        T f = (T)1;
        if (u > (T)0) {
            const T delta = (T)((eps_new * eps_new - eps_old * eps_old) / (1.0 - eps_new * eps_new)) *
                            (T)wmag[j] / u;
            const T a = sqrt((T)((1.0 - eps_old * eps_old)) /
                             ((T)(1.0 - eps_new * eps_new) * ((T)1 + delta)));
            f = (T)0.001 * floor((T)1000 * a);
        }
        w[i] *= f;
    }
}

// ws /= 10*sqrt(uj) (:117)
template <typename T>
__global__ void init_scale_kernel(T* __restrict__ w, int64_t n, int Mp, int m,
                                  const double* __restrict__ uj) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Mp);
        if (j < m) w[i] = w[i] / ((T)10 * sqrt((T)uj[j]));
    }
}

// ws[order] (:162): out[v][j] = in[v][order[j]] for j < m
template <typename T>
__global__ void permute_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n, int Mp,
                               int m, const int* __restrict__ order) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Mp);
        out[i] = (j < m) ? in[i - j + order[j]] : (T)0;
    }
}

// ------------------------------------------------------------------------------------------------
// detail moments (:277-287)
// ------------------------------------------------------------------------------------------------
// inverse of ry by Gauss-Jordan with partial pivoting; one block; work: [Mp][2*Mp] doubles (global)
static __global__ void invert_kernel(const double* __restrict__ a, int Mp, double* __restrict__ work,
                              double* __restrict__ inv) {
    __shared__ int piv_s;
    __shared__ double pval_s;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int W2 = 2 * Mp;
    for (int idx = tid; idx < Mp * W2; idx += nt) {
        const int r = idx / W2, c = idx % W2;
        work[idx] = (c < Mp) ? a[r * Mp + c] : ((c - Mp == r) ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int col = 0; col < Mp; ++col) {
        if (tid == 0) {
            int p = col;
            double best = fabs(work[col * W2 + col]);
            for (int r = col + 1; r < Mp; ++r) {
                const double x = fabs(work[r * W2 + col]);
                if (x > best) { best = x; p = r; }
            }
            piv_s = p;
        }
        __syncthreads();
        const int p = piv_s;
        if (p != col) {
            for (int c = tid; c < W2; c += nt) {
                const double t = work[col * W2 + c];
                work[col * W2 + c] = work[p * W2 + c];
                work[p * W2 + c] = t;
            }
        }
        __syncthreads();
        if (tid == 0) pval_s = work[col * W2 + col];
        __syncthreads();
        const double pv = pval_s;
        for (int c = tid; c < W2; c += nt) work[col * W2 + c] /= pv;
        __syncthreads();
        for (int idx = tid; idx < Mp * W2; idx += nt) {
            const int r = idx / W2, c = idx % W2;
            if (r != col && c != col) work[idx] -= work[r * W2 + col] * work[col * W2 + c];
        }
        __syncthreads();
        for (int r = tid; r < Mp; r += nt)
            if (r != col) work[r * W2 + col] = 0.0;
        __syncthreads();
    }
    for (int idx = tid; idx < Mp * Mp; idx += nt) inv[idx] = work[(idx / Mp) * W2 + Mp + idx % Mp];
}

// X_i Z_j = solve(ry, rho)^T (:280), X_i^2|Y (:281), MI (:278) and the sums behind TCs,
// TC_no_overlap, TC_direct, additivity (:284-287).
// partial sums per block: [0..m) sum_i MI_ji ; [m] sum_i max_j MI ; [m+1] sum_i I(X_i;Y) ; [m+2] sum_ij MI
// optional outputs (may be null): mi_o, xz_o [Vp][Mp], x2y_o [Vp]
// dynamic LDS: ri_s[Mp*Mp] (T) + rho_s[VPB*Mp] (T) + acc_s[VPB*Mp] (double)
template <typename T, int Mp>
__global__ void __launch_bounds__(Pvt<Mp>::v)
detail_kernel(const T* __restrict__ rho_i, const double* __restrict__ ryinv, int64_t V, int m,
              T* __restrict__ mi_o, T* __restrict__ xz_o, T* __restrict__ x2y_o,
              double* __restrict__ dpart_sums, const double* __restrict__ xz_fscale = nullptr,
              T* __restrict__ inv_x2y_o = nullptr) {
    constexpr int NT = Pvt<Mp>::v;
    constexpr int VPB = NT / Mp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* ri_s = reinterpret_cast<T*>(smem_raw);
    T* rho_s = ri_s + (OpInLds<Mp>::v ? Mp * Mp : 0);
    double* acc_s = reinterpret_cast<double*>(rho_s + VPB * Mp + (((VPB * Mp) & 1) ? 1 : 0));
    __shared__ T gs_scratch[NT / 64];
    __shared__ double bs_scratch[NT / 64];
    const int tid = threadIdx.x, vl = tid / Mp, j = tid % Mp;
    if (OpInLds<Mp>::v)
        for (int idx = tid; idx < Mp * Mp; idx += NT) ri_s[idx] = (T)ryinv[idx];
    __syncthreads();
    double col_mi = 0.0, s_max = 0.0, s_ixy = 0.0;
    const int64_t ngroups = (V + VPB - 1) / VPB;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t v = grp * VPB + vl;
        const bool ok = v < V;
        const int64_t o = (ok ? v : 0) * Mp + j;
        const T rho = ok ? rho_i[o] : (T)0;
        __syncthreads();
        rho_s[vl * Mp + j] = rho;
        __syncthreads();
        T xz = (T)0;
        if (OpInLds<Mp>::v) {
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) xz += ri_s[j * Mp + k] * rho_s[vl * Mp + k];   // (ry^-1 rho)_j
        } else {
            // ry^-1 is symmetric up to rounding: column j, coalesced over j
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) xz += (T)ryinv[k * Mp + j] * rho_s[vl * Mp + k];
        }
        const T mi = (T)-0.5 * log1p(-rho * rho);
        const T dot = group_sum<Mp, T>(xz * rho, gs_scratch, tid);
        // max over factors of MI (padded factors have MI = 0 <= every real MI)
        T mx = mi;
        {
            constexpr int Wd = Mp < 64 ? Mp : 64;
#pragma unroll
            for (int off = Wd / 2; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, Wd));
            if (Mp > 64) {
                constexpr int NW = Mp / 64;
                __syncthreads();
                if ((tid & 63) == 0) gs_scratch[tid >> 6] = mx;
                __syncthreads();
                const int base = ((tid >> 6) / NW) * NW;
                mx = gs_scratch[base];
#pragma unroll
                for (int w = 1; w < NW; ++w) mx = fmax(mx, gs_scratch[base + w]);
            }
        }
        T x2y = (T)1 - dot;
        x2y = x2y < (T)1e-6 ? (T)1e-6 : x2y;                                       // clip, :281
        if (ok) {
            if (mi_o) mi_o[o] = mi;
            if (xz_o) xz_o[o] = xz_fscale ? xz * (T)xz_fscale[j] : xz;
            if (j < m) col_mi += (double)mi;
            if (j == 0) {
                if (x2y_o) x2y_o[v] = x2y;
                if (inv_x2y_o) inv_x2y_o[v] = (T)1 / x2y;
                s_max += (double)mx;
                s_ixy += (double)((T)-0.5 * log(x2y));
            }
        }
    }
    acc_s[vl * Mp + j] = col_mi;
    __syncthreads();
    double* outp = dpart_sums + (int64_t)blockIdx.x * (m + 3);      // compact: m sums + 3 scalars
    double colsum = 0.0;
    if (tid < m) {
        for (int k = 0; k < VPB; ++k) colsum += acc_s[k * Mp + tid];
        outp[tid] = colsum;
    }
    const double tot = block_sum<double, NT>(tid < m ? colsum : 0.0, bs_scratch, tid);
    s_max = block_sum<double, NT>(s_max, bs_scratch, tid);
    s_ixy = block_sum<double, NT>(s_ixy, bs_scratch, tid);
    if (tid == 0) { outp[m] = s_max; outp[m + 1] = s_ixy; outp[m + 2] = tot; }
}

// ------------------------------------------------------------------------------------------------
// synergistic branch (discourage_overlap=False; linearcorex.py:336-384)
// ------------------------------------------------------------------------------------------------
// per-factor part of _calculate_moments_syn from the Gram of the (all-reduced) Y:
//   cy = Y^T Y / N + yscale^2 I  (== ws.dot(X_i Y_j) + yscale^2 I, :356, because W X^T = Y^T)
//   Y_j^2 = diag(cy) (:357), ry = cy / (sd_j sd_k) (:358), 1/sd_j, sum_j I(Y_j;X) (:369)
// one block; gy: [nsplit][Mp][Mp] partials
template <typename T>
__global__ void __launch_bounds__(256)
syn_small_kernel(const T* __restrict__ gy, int nsplit, int Mp, int m, double n_samples, double yscale,
                 double* __restrict__ cy, double* __restrict__ yj2, double* __restrict__ ry,
                 double* __restrict__ inv_sd, SetState* st) {
    const int tid = threadIdx.x;
    const int mm = Mp * Mp;
    for (int idx = tid; idx < mm; idx += blockDim.x) {
        T g = (T)0;
        for (int k = 0; k < nsplit; ++k) g += gy[(int64_t)k * mm + idx];
        const int a = idx / Mp, b = idx % Mp;
        T c = g / (T)n_samples;
        if (a == b) c += (T)(yscale * yscale);
        cy[idx] = (double)c;
    }
    __syncthreads();
    for (int j = tid; j < Mp; j += blockDim.x) {            // (any number of factors: the wide path has more than one per thread)
        const double d = cy[j * Mp + j];
        yj2[j] = d;
        inv_sd[j] = 1.0 / sqrt((double)(T)d);
    }
    __syncthreads();
    for (int idx = tid; idx < mm; idx += blockDim.x) {
        const int a = idx / Mp, b = idx % Mp;
        ry[idx] = (double)((T)cy[idx] / ((T)sqrt((double)(T)yj2[a]) * (T)sqrt((double)(T)yj2[b])));
    }
    if (tid == 0) {
        // the reference keeps W and every moment of this branch in float64 (:121), also when x is float32
        double s = 0.0;
        for (int j = 0; j < m; ++j) s += 0.5 * log(yj2[j]) - 0.5 * log(yscale * yscale);
        st->sum_log_rj = s;           // here: sum_j I(Y_j ; X)
        st->max_uj = 0.0;
        st->invalid = 0;
        st->invalid_d = 0.0;
    }
}

// D = X^T Y (sum of the partial slots) and rho_ji = <X_i Y_j> / sd_j (:355, :359)
template <typename T>
__global__ void syn_rho_kernel(const T* __restrict__ dpart, int nsplit, int64_t pstride, int64_t total, int Mp,
                               double n_samples, const double* __restrict__ inv_sd, T* __restrict__ d_out,
                               T* __restrict__ rho_o) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        T d = dpart[o];
        for (int k = 1; k < nsplit; ++k) d += dpart[k * pstride + o];
        d_out[o] = d;
        rho_o[o] = d / (T)n_samples * (T)inv_sd[o % Mp];
    }
}

// TC = sum_i I(X_i;Y) - sum_j I(Y_j;X) (:373); publish
template <typename T>
__global__ void syn_tc_kernel(const double* __restrict__ sbuf, int m, SetState* st, SetState* host, unsigned int seq) {
    st->tc = sbuf[m + 1] - st->sum_log_rj;      // float64 like the reference's synergistic moments (:121, :373)
    publish_state(st, host, seq);
}

// _update_syn (:375-383): ws' = (1-eta) ws + eta (X_i Z_j^T / X_i^2|Y - H ws), H diagonal zeroed
// dynamic LDS: h_s[Mp*(Mp+1)] (T) + w_s[VPB*Mp] (T)
template <typename T, int Mp>
__global__ void __launch_bounds__(Pvt<Mp>::v)
syn_update_kernel(const T* __restrict__ W, const T* __restrict__ xz, const T* __restrict__ inv_x2y,
                  const double* __restrict__ H, int64_t V, T eta, T* __restrict__ w_out) {
    constexpr int NT = Pvt<Mp>::v;
    constexpr int VPB = NT / Mp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* h_s = reinterpret_cast<T*>(smem_raw);
    T* w_s = h_s + (OpInLds<Mp>::v ? Mp * (Mp + 1) : 0);
    const int tid = threadIdx.x, vl = tid / Mp, j = tid % Mp;
    if (OpInLds<Mp>::v)
        for (int idx = tid; idx < Mp * Mp; idx += NT) {
            const int a = idx / Mp, b = idx % Mp;
            h_s[a * (Mp + 1) + b] = (a == b) ? (T)0 : (T)H[idx];
        }
    __syncthreads();
    const int64_t ngroups = (V + VPB - 1) / VPB;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t v = grp * VPB + vl;
        const bool ok = v < V;
        const int64_t o = (ok ? v : 0) * Mp + j;
        const T w = ok ? W[o] : (T)0;
        __syncthreads();
        w_s[vl * Mp + j] = w;
        __syncthreads();
        T s = (T)0;
        if (OpInLds<Mp>::v) {
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) s += h_s[j * (Mp + 1) + k] * w_s[vl * Mp + k];
        } else {
#pragma unroll 8
            for (int k = 0; k < Mp; ++k) s += (k == j ? (T)0 : (T)H[k * Mp + j]) * w_s[vl * Mp + k];
        }
        if (ok) {
            const T r = xz[o] * inv_x2y[v];
            w_out[o] = ((T)1 - eta) * w + eta * (r - s);
        }
    }
}

// derived arrays for readback
template <typename T>
__global__ void invrho_kernel(const T* __restrict__ rho, int64_t n, T* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (T)1 / ((T)1 - rho[i] * rho[i]);
}

// ------------------------------------------------------------------------------------------------
// get_covariance (:443-455): a rank-Mp product of two [V][Mp] operands, written once.
//   non-synergistic (:446-451): out[r][c] = std_r std_c (r == c ? 1 : z_r . z_c / (1 - eps^2)),  z_i = rhoinvrho_i / (1 + Si_i)
//   synergistic     (:452-455): out[r][c] = std_r std_c (r == c ? 1 : X_i Z_j[r] . X_i Y_j[c]),  X_i Y_j = X^T.Y / N
// cov_prep_kernel forms the operand(s) once per call (elementwise, [Vp][Mp]); cov_syrk_kernel is the product: one
// 64 x 64 output tile per block of 4 waves, MFMA 16x16x4, operands staged through LDS in chunks of <= 32 factors
// (rows padded by 2 elements: the 16 rows x 2 contraction lanes of a read land on distinct banks in both precisions).
// The kernel is bound by the V^2 output write: a lane owns 4 CONSECUTIVE columns (MFMA column j of tile u is column
// 4 j + u of the block), so every store instruction writes 16 lanes x 16 / 32 bytes = whole 256 / 512-byte row segments.
// The output block has a padded leading dimension (ldo, a multiple of 64): columns past V land in the padding.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cov_prep_kernel(const T* __restrict__ src_a, const T* __restrict__ si, const T* __restrict__ src_b,
                                int64_t n /* Vp * Mp */, int Mp, T inv_n, T* __restrict__ op_a, T* __restrict__ op_b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (si != nullptr) op_a[i] = src_a[i] / ((T)1 + si[i / Mp]);       // z = rhoinvrho / (1 + Si), :447
        if (src_b != nullptr) op_b[i] = src_b[i] * inv_n;                  // X_i Y_j = X^T.Y / N, :355
    }
}

// PREDICT false: get_covariance (rows are variables, symmetric scaling std_r std_c, unit diagonal).
// PREDICT true:  predict (:440-441) - rows are the n_rows samples of a staged block of Y (A = Y [rows][Mp], B = X_i Z_j [V][Mp]),
//         and the epilogue is `invert` (:431-438): out[r][c] = std_c f(y_r . xz_c) + mean_c with f = identity ('standard',
//         kind 1) or g_inv (:490-494; 'outliers', kind 2); kind 0 writes the product unchanged.  aux = mean.
template <typename T>
__device__ __forceinline__ T g_inv_dev(T x) {
    const T t = (T)4;
    const T xp = x < -t ? -t : (x > t ? t : x);
    const T lo = (T)(-1 + 1e-10), hi = (T)(1 - 1e-10);          // as the reference clips, in the working precision
    T d = x - xp;
    d = d < lo ? lo : (d > hi ? hi : d);
    return xp + (T)atanh((double)d);
}

template <typename T, int Mp, bool PREDICT>
__global__ void __launch_bounds__(256)
cov_syrk_kernel(const T* __restrict__ A, const T* __restrict__ B, const T* __restrict__ stdv, int64_t V, int64_t row0,
                int64_t nrows, T denom, T* __restrict__ out, int64_t ldo, const T* __restrict__ aux, int kind) {
    constexpr int KC = Mp < 32 ? Mp : 32;
    constexpr int LDS_LD = KC + 2;
    typedef typename MF<T>::acc_t acc_t;
    __shared__ T As[64][LDS_LD];
    __shared__ T Bs[64][LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t rb = row0 + (int64_t)blockIdx.y * 64, cb = (int64_t)blockIdx.x * 64;
    acc_t acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = (acc_t){0, 0, 0, 0};
    for (int j0 = 0; j0 < Mp; j0 += KC) {
        __syncthreads();
        for (int idx = tid; idx < 64 * KC; idx += 256) {
            const int a = idx / KC, j = idx % KC;
            const int64_t vr = rb + a, vc = cb + a;
            As[a][j] = ((PREDICT || vr < V) && vr < row0 + nrows) ? A[vr * Mp + j0 + j] : (T)0;
            Bs[(a & 3) * 16 + (a >> 2)][j] = (vc < V) ? B[vc * Mp + j0 + j] : (T)0;      // column 4 j + u -> row 16 u + j
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const T av = As[16 * wave + i][4 * s + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = MF<T>::mma(av, Bs[16 * u + i][4 * s + q], acc[u]);
        }
    }
    Pk<T, 4> sc, mu;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t c = cb + 4 * i + u;
        sc.v[u] = (c < V && (!PREDICT || kind != 0)) ? stdv[c] : (T)(PREDICT ? 1 : 0);
        mu.v[u] = (PREDICT && kind != 0 && c < V) ? aux[c] : (T)0;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t r = rb + 16 * wave + MF<T>::row(lane, g);
        if ((!PREDICT && r >= V) || r >= row0 + nrows) continue;
        Pk<T, 4> o;
        if (!PREDICT) {
            const T sr = stdv[r];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t c = cb + 4 * i + u;
                const T val = (r == c) ? (T)1 : acc[u][g] / denom;
                o.v[u] = sr * sc.v[u] * val;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T val = (T)acc[u][g];
                o.v[u] = sc.v[u] * (kind == 2 ? g_inv_dev<T>(val) : val) + mu.v[u];
            }
        }
        *reinterpret_cast<Pk<T, 4>*>(out + (r - row0) * ldo + cb + 4 * i) = o;
    }
}

// invert (:431-438) of a staged block of rows, elementwise: out = std_c f(x) + mean_c (f as in the predict epilogue of cov_syrk_kernel)
template <typename T>
__global__ void invert_rows_kernel(const T* __restrict__ x, int64_t nrows, int64_t V, int64_t ld, const T* __restrict__ mean,
                                   const T* __restrict__ stdv, int kind, T* __restrict__ out) {
    const int64_t total = nrows * V;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = k / V, c = k % V;
        const T v = x[r * ld + c];
        out[r * ld + c] = kind == 0 ? v : stdv[c] * (kind == 2 ? g_inv_dev<T>(v) : v) + mean[c];
    }
}

// ------------------------------------------------------------------------------------------------
// second resident copy of the shard, transposed: XT[v][n] = X[n][v]  ([Vp][Npad], zero padded).
// With 288 GB of HBM both layouts fit, and both X-streaming contractions then read their big
// operand with 16 consecutive lanes on 256 contiguous bytes (gemm_tn's pattern).
// 64x64 tiles through LDS; grid = (Vp/64, Npad/64), 256 threads.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
transpose_kernel(const T* __restrict__ X, int64_t ldx, T* __restrict__ XT, int64_t ldt) {
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t v0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = X[(n0 + r) * ldx + v0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) XT[(v0 + r) * ldt + n0 + tx] = tile[tx][r];
}

// ------------------------------------------------------------------------------------------------
// panel-major resident shard (gemm_kernels.hpp, PanelW): a column block staged row-major S[rows][ld] <-> its panels
// XP[(col0 + c) / PW][n][PW].  One thread moves 16 bytes; consecutive threads walk a panel row, then the rows of the panel: the panel
// side is fully contiguous, the staging side in 64-byte runs.  TO_PANEL = false copies back (lcx_download_x).
// ------------------------------------------------------------------------------------------------
template <typename T, bool TO_PANEL>
__global__ void __launch_bounds__(256)
panel_block_kernel(T* __restrict__ S, int64_t ld, T* __restrict__ XP, int64_t panel_stride, int64_t rows, int64_t col0, int64_t ncols) {
    constexpr int PW = 64 / (int)sizeof(T), E = 16 / (int)sizeof(T), QP = PW / E;       // 4 pieces of 16 B per panel row
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int64_t npan = ncols / PW;                                                     // ncols, col0: multiples of PW
    const int64_t total = npan * rows * QP;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pn = k / (rows * QP), rem = k - pn * rows * QP;
        const int64_t n = rem / QP;
        const int c = (int)(rem % QP);
        f4* pp = reinterpret_cast<f4*>(XP + (col0 / PW + pn) * panel_stride + n * PW + c * E);
        f4* sp = reinterpret_cast<f4*>(S + n * ld + pn * PW + c * E);
        if (TO_PANEL) *pp = *sp; else *sp = *pp;
    }
}

// ------------------------------------------------------------------------------------------------
// synthetic data on device: counter-based N(0,1), keyed by (seed, row, global col)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double gauss_at(uint64_t seed, uint64_t a, uint64_t b) {
    const uint64_t h1 = mix64(seed ^ mix64(a * 0xD1342543DE82EF95ull + b));
    const uint64_t h2 = mix64(h1 ^ 0xA0761D6478BD642Full);
    const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740993.0);
    const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

template <typename T>
__global__ void generate_kernel(T* __restrict__ X, int64_t N, int64_t V, int64_t ldx, uint64_t seed,
                                int kind, int n_groups, int64_t col_offset) {
    const int64_t total = N * V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / V, c = i % V, gc = c + col_offset;
        double x = gauss_at(seed, (uint64_t)r, (uint64_t)gc);
        if (kind == 1) {
            const uint64_t grp = mix64(seed * 31 + (uint64_t)gc) % (uint64_t)n_groups;
            x += gauss_at(seed ^ 0x5851F42D4C957F2Dull, (uint64_t)r, (1ull << 40) + grp);
        }
        X[r * ldx + c] = (T)x;
    }
}

// ------------------------------------------------------------------------------------------------
// preprocess on device (reference linearcorex.py:397-429, mean_impute :497-510, g :483-487).
// Three coalesced passes over the resident shard, 64-column strips x row splits, double accumulators:
//   A  observed count and sum per column          -> n_obs, imputation mean
//   B  sum of (x - mean)^2 over observed cells    -> std  ('standard': / n_obs, 'outliers': / N; clip 1e-10)
//   C  impute, (x - mean) / std, optional tail squash g, in place; max |x~| per block
// A cell is missing if it is NaN or equals the sentinel (only when missing values are enabled, as in
// the reference); infinities are neither observed nor imputed (:505-507).
// ------------------------------------------------------------------------------------------------
constexpr int PP_KIND_NONE = 0, PP_KIND_STANDARD = 1, PP_KIND_OUTLIERS = 2, PP_KIND_EMPIRICAL = 3;   // 3: empirical.hip

template <typename T>
__device__ __forceinline__ bool pp_missing(T x, int has_missing, T sentinel) {
    return has_missing && (x != x || x == sentinel);
}

// pass A (center == nullptr): part_s = sum of observed, part_n = count of observed
// pass B (center != nullptr): part_s = sum of (x - center)^2 over observed cells
template <typename T>
__global__ void __launch_bounds__(256)
pp_colsum_kernel(const T* __restrict__ X, int64_t N, int64_t V, int64_t ldx, int has_missing, T sentinel,
                 const double* __restrict__ center, double* __restrict__ part_s, double* __restrict__ part_n) {
    __shared__ double sh_s[4][64];
    __shared__ double sh_n[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int RS = gridDim.y;
    const int64_t r0 = N * blockIdx.y / RS, r1 = N * (blockIdx.y + 1) / RS;
    double s = 0.0, n = 0.0;
    if (col < V) {
        const double mu = center ? center[col] : 0.0;
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const T x = X[r * ldx + col];
            const bool obs = has_missing ? (!pp_missing(x, 1, sentinel) && isfinite((double)x)) : true;
            if (obs) {
                if (center) { const double d = (double)x - mu; s += d * d; }
                else { s += (double)x; n += 1.0; }
            }
        }
    }
    sh_s[rl][c] = s;
    sh_n[rl][c] = n;
    __syncthreads();
    if (rl == 0 && col < V) {
        part_s[(int64_t)blockIdx.y * V + col] = sh_s[0][c] + sh_s[1][c] + sh_s[2][c] + sh_s[3][c];
        if (part_n) part_n[(int64_t)blockIdx.y * V + col] = sh_n[0][c] + sh_n[1][c] + sh_n[2][c] + sh_n[3][c];
    }
}

// after pass A: nobs, imputation mean (= mean of the observed cells); after pass B: std
template <typename T>
__global__ void pp_finalize_kernel(const double* __restrict__ part_s, const double* __restrict__ part_n, int RS,
                                   int64_t V, double n_rows, int kind, int pass, double* __restrict__ nobs,
                                   double* __restrict__ mean, double* __restrict__ stdv) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= V) return;
    double s = 0.0, n = 0.0;
    for (int k = 0; k < RS; ++k) {
        s += part_s[(int64_t)k * V + c];
        if (part_n) n += part_n[(int64_t)k * V + c];
    }
    if (pass == 0) {
        nobs[c] = n;
        mean[c] = (double)(T)(s / n);                       // the reference holds theta in the working dtype
    } else {
        const double denom = kind == PP_KIND_OUTLIERS ? n_rows : nobs[c];
        double sd = (double)(T)sqrt(s / denom);
        if (sd < 1e-10) sd = 1e-10;
        stdv[c] = sd;
    }
}

template <typename T>
__device__ __forceinline__ T pp_g(T x) {                      // :483-487, t = 4
    const T core = x < (T)-4 ? (T)-4 : (x > (T)4 ? (T)4 : x);
    return core + tanh(x - core);
}

// pass C, in place.  impute[c] = mean of the observed cells of THIS data (what mean_impute writes),
// mean/stdv = theta (of the fitted data).  blockmax[b] = max |x~| seen by block b.
template <typename T>
__global__ void __launch_bounds__(256)
pp_apply_kernel(T* __restrict__ X, int64_t N, int64_t V, int64_t ldx, int has_missing, T sentinel,
                const double* __restrict__ impute, const double* __restrict__ mean,
                const double* __restrict__ stdv, int kind, double* __restrict__ blockmax) {
    __shared__ double sh[4];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int RS = gridDim.y;
    const int64_t r0 = N * blockIdx.y / RS, r1 = N * (blockIdx.y + 1) / RS;
    double mx = 0.0;
    if (col < V) {
        const T mu = kind == PP_KIND_NONE ? (T)0 : (T)mean[col];
        const T sd = kind == PP_KIND_NONE ? (T)1 : (T)stdv[col];
        const T imp = has_missing ? (T)impute[col] : (T)0;
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            T x = X[r * ldx + col];
            if (pp_missing(x, has_missing, sentinel)) x = imp;
            if (kind != PP_KIND_NONE) {
                x = (x - mu) / sd;
                if (kind == PP_KIND_OUTLIERS) x = pp_g(x);
            }
            X[r * ldx + col] = x;
            const double ax = fabs((double)x);
            mx = ax > mx ? ax : mx;                             // NaN never wins, like np.max would propagate - only used for a warning
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0) sh[rl] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = sh[0];
        for (int k = 1; k < 4; ++k) m = sh[k] > m ? sh[k] : m;
        blockmax[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
}


// ------------------------------------------------------------------------------------------------
// The two per-variable kernels with an m x m operator - moments_epilogue (Qij = ry . rhoinvrho, :266) and grad (H0 . ws, :300) - with
// that product on the MATRIX pipe (float32, 64 / 128 padded factors: the large-shard configurations).  The thread-per-(variable,
// factor) forms above walk the operator out of LDS once per variable: 2 LDS operands per FMA, LDS-issue bound (128 factors:
// 555 / 337 us for 450 MB of traffic, profiles/r04_trace_gaps_c4shard*.txt).  Here a WAVE owns 16 consecutive variables:
//   * elementwise phases in the FLAT layout - the 16 x Mp block of a [V][Mp] array is 16 Mp contiguous floats, lane l takes the
//     float4 pieces r*64 + l (r < Mp/16): every load / store instruction covers 1 KB contiguous; a variable's Mp factors sit in
//     Mp/4 = 32 / 16 consecutive lanes, so the per-variable sums (Si, Qi-Si^2, :268-269) are xor-shuffles inside the wave;
//   * the operator product as 16 x 16 x 4 float32 MFMAs: A = the wave's 16 x Mp tile (through the wave's own LDS tile: written
//     flat, read as 4 consecutive contraction elements per lane), B = the operator, staged once per block [Mp][Mp + 4] (the
//     4 contraction rows of a step land on distinct banks), D back through the same LDS tile into the flat layout;
//     Mp^2 / 64 MFMAs per 16 variables: 30 us of matrix pipe for 125 000 x 128.
// Same outputs, same per-block partial sums (tcpart / bjpart, one slot per block of the same pv_grid) as the kernels above; the
// operator product rounds in another order (MFMA chain over k instead of a serial loop).
// ------------------------------------------------------------------------------------------------
template <int Mp> struct PvMfma {
    static constexpr int NW = 8;                    // waves per block
    static constexpr int LD = Mp + 4;               // padded LDS row (floats)
    static constexpr int F4 = Mp / 16;              // float4 pieces per lane and 16-variable group
    static constexpr int LPV = Mp / 4;              // lanes per variable in the flat layout
    static constexpr size_t lds_bytes = ((size_t)Mp * LD + (size_t)NW * 16 * LD) * sizeof(float);
};
typedef float pv_f4 __attribute__((ext_vector_type(4)));

// sum over the LPV consecutive lanes that hold one variable
template <int LPV>
__device__ __forceinline__ float lanes_sum(float v) {
#pragma unroll
    for (int off = LPV / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// D tile (16 variables x Mp) = A tile (in `tile`, flat rows of LD floats) . op_s (Mp x Mp, rows of LD floats, B[k][j] = op_s[k*LD + j]);
// the result replaces the A tile.  One wave; LDS operations of a wave execute in order.
template <int Mp>
__device__ __forceinline__ void pv_tile_product(float* tile, const float* op_s, int lane) {
    constexpr int LD = PvMfma<Mp>::LD, NT16 = Mp / 16;
    typedef float acc_t __attribute__((ext_vector_type(4)));
    const int i = lane & 15, kq = lane >> 4;
    pv_f4 a[NT16];
#pragma unroll
    for (int S = 0; S < NT16; ++S) a[S] = *reinterpret_cast<const pv_f4*>(&tile[i * LD + 16 * S + 4 * kq]);
    acc_t acc[NT16];
#pragma unroll
    for (int u = 0; u < NT16; ++u) acc[u] = (acc_t){0, 0, 0, 0};
#pragma unroll
    for (int S = 0; S < NT16; ++S)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* brow = op_s + (16 * S + 4 * kq + e) * LD + i;
#pragma unroll
            for (int u = 0; u < NT16; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S][e], brow[16 * u], acc[u], 0, 0, 0);
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < NT16; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * kq + r) * LD + 16 * u + i] = acc[u][r];
    __builtin_amdgcn_wave_barrier();
}

template <int Mp>
__global__ void __launch_bounds__(64 * PvMfma<Mp>::NW)
moments_epilogue_mfma_kernel(const float* __restrict__ dpart, int nsplit, int64_t pstride, const float* __restrict__ d_base,
                             const float* __restrict__ d_dir, float eta, float* __restrict__ d_out, const float* __restrict__ W,
                             const double* __restrict__ ry, int64_t V, double n_samples, double eps, float* __restrict__ rho_o,
                             float* __restrict__ rir_o, float* __restrict__ qij_o, float* __restrict__ si_o, float* __restrict__ q2_o,
                             float* __restrict__ hscale_o, double* __restrict__ tcpart, const int* __restrict__ skip_flag) {
    constexpr int NW = PvMfma<Mp>::NW, LD = PvMfma<Mp>::LD, F4 = PvMfma<Mp>::F4, LPV = PvMfma<Mp>::LPV, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ry_s = reinterpret_cast<float*>(smem_raw);
    __shared__ double bs_scratch[NW];
    if (skip_flag != nullptr && *skip_flag != 0) return;       // invalid trial (:250-251): the tail block still publishes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* tile = ry_s + Mp * LD + wave * 16 * LD;
    for (int idx = tid; idx < Mp * Mp; idx += NTH) ry_s[(idx / Mp) * LD + idx % Mp] = (float)ry[idx];     // symmetric
    __syncthreads();

    const float c1 = (float)(1.0 - eps * eps), c2 = (float)(eps * eps), ns = (float)n_samples;
    double s1 = 0.0, s2 = 0.0;
    const int64_t ngroups = (V + 15) / 16;
    for (int64_t grp = (int64_t)blockIdx.x * NW + wave; grp < ngroups; grp += (int64_t)gridDim.x * NW) {
        const int64_t base4 = grp * (16 * Mp / 4);
        pv_f4 rho[F4], rir[F4];
        float si[F4];
#pragma unroll
        for (int r = 0; r < F4; ++r) {
            const int f = r * 64 + lane, vloc = (4 * f) / Mp, j0 = (4 * f) % Mp;
            const bool ok = grp * 16 + vloc < V;
            pv_f4 d;
            if (d_base != nullptr) {
                d = reinterpret_cast<const pv_f4*>(d_base)[base4 + f] + eta * reinterpret_cast<const pv_f4*>(d_dir)[base4 + f];
            } else {
                d = reinterpret_cast<const pv_f4*>(dpart)[base4 + f];
                for (int k = 1; k < nsplit; ++k) d += reinterpret_cast<const pv_f4*>(dpart + k * pstride)[base4 + f];
            }
            if (ok) reinterpret_cast<pv_f4*>(d_out)[base4 + f] = d;
            const pv_f4 w = reinterpret_cast<const pv_f4*>(W)[base4 + f];
            pv_f4 rh, rr;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                rh[c] = ok ? (c1 * d[c] / ns + c2 * w[c]) : 0.f;
                rr[c] = rh[c] * (1.f / (1.f - rh[c] * rh[c]));
            }
            rho[r] = rh;
            rir[r] = rr;
            si[r] = lanes_sum<LPV>(rh[0] * rr[0] + rh[1] * rr[1] + rh[2] * rr[2] + rh[3] * rr[3]);
            *reinterpret_cast<pv_f4*>(&tile[vloc * LD + j0]) = rr;
        }
        __builtin_amdgcn_wave_barrier();
        pv_tile_product<Mp>(tile, ry_s, lane);                 // Qij = ry . rhoinvrho (:266)
#pragma unroll
        for (int r = 0; r < F4; ++r) {
            const int f = r * 64 + lane, vloc = (4 * f) / Mp, j0 = (4 * f) % Mp;
            const int64_t v = grp * 16 + vloc;
            const bool ok = v < V;
            const pv_f4 q = *reinterpret_cast<const pv_f4*>(&tile[vloc * LD + j0]);
            float p = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) p += rir[r][c] * (q[c] - si[r] * rho[r][c]);
            const float q2 = lanes_sum<LPV>(p);
            if (ok) {
                reinterpret_cast<pv_f4*>(rho_o)[base4 + f] = rho[r];
                reinterpret_cast<pv_f4*>(rir_o)[base4 + f] = rir[r];
                reinterpret_cast<pv_f4*>(qij_o)[base4 + f] = q;
                if ((lane & (LPV - 1)) == 0) {
                    si_o[v] = si[r];
                    q2_o[v] = q2;
                    hscale_o[v] = 1.f / (1.f + q2);
                    s1 += (double)logf(1.f + si[r]);
                    s2 += (double)logf(1.f + q2);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                       // the tile is rewritten by the next group
    }
    s1 = block_sum<double, NTH>(s1, bs_scratch, tid);
    s2 = block_sum<double, NTH>(s2, bs_scratch, tid);
    if (tid == 0) {
        tcpart[2 * blockIdx.x] = s1;
        tcpart[2 * blockIdx.x + 1] = s2;
    }
}

// grad (:296-300) + the per-block Bj partials (:302); grad2_o: columns [0, Mp) of the merged operand [V][2 Mp]
template <int Mp>
__global__ void __launch_bounds__(64 * PvMfma<Mp>::NW)
grad_mfma_kernel(const float* __restrict__ W, const float* __restrict__ rho_i, const float* __restrict__ rir_i,
                 const float* __restrict__ qij_i, const float* __restrict__ si_i, const float* __restrict__ q2_i,
                 const double* __restrict__ uj, const double* __restrict__ H /* sbuf, diag ignored */, int64_t V,
                 float* __restrict__ grad_o, double* __restrict__ bjpart, float* __restrict__ grad2_o) {
    constexpr int NW = PvMfma<Mp>::NW, LD = PvMfma<Mp>::LD, F4 = PvMfma<Mp>::F4, LPV = PvMfma<Mp>::LPV, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* h_s = reinterpret_cast<float*>(smem_raw);           // h_s[k][j] = H0[j][k]: the B operand of hw[v][j] = sum_k H0[j][k] w[v][k]
    __shared__ double bj_s[NW][Mp];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* tile = h_s + Mp * LD + wave * 16 * LD;
    for (int idx = tid; idx < Mp * Mp; idx += NTH) {
        const int a = idx / Mp, b = idx % Mp;                  // H[a][b]
        h_s[b * LD + a] = (a == b) ? 0.f : (float)H[idx];      // fill_diagonal(H, 0), :295
    }
    __syncthreads();
    const int j0 = (4 * lane) % Mp;                            // the 4 factors of this lane (the same for every piece r)
    float rj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) rj[c] = 1.f - (float)uj[j0 + c];
    double bj[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t ngroups = (V + 15) / 16;
    for (int64_t grp = (int64_t)blockIdx.x * NW + wave; grp < ngroups; grp += (int64_t)gridDim.x * NW) {
        const int64_t base4 = grp * (16 * Mp / 4);
        pv_f4 w[F4];
#pragma unroll
        for (int r = 0; r < F4; ++r) {
            const int f = r * 64 + lane, vloc = (4 * f) / Mp;
            const bool ok = grp * 16 + vloc < V;
            w[r] = reinterpret_cast<const pv_f4*>(W)[base4 + f];
            if (!ok) w[r] = (pv_f4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<pv_f4*>(&tile[vloc * LD + j0]) = w[r];
        }
        __builtin_amdgcn_wave_barrier();
        pv_tile_product<Mp>(tile, h_s, lane);                  // H0 . ws (:300)
#pragma unroll
        for (int r = 0; r < F4; ++r) {
            const int f = r * 64 + lane, vloc = (4 * f) / Mp;
            const int64_t v = grp * 16 + vloc;
            const bool ok = v < V;
            const int64_t vv = ok ? v : 0;
            const pv_f4 hw = *reinterpret_cast<const pv_f4*>(&tile[vloc * LD + j0]);
            const pv_f4 rho = reinterpret_cast<const pv_f4*>(rho_i)[base4 + f], rir = reinterpret_cast<const pv_f4*>(rir_i)[base4 + f],
                        qij = reinterpret_cast<const pv_f4*>(qij_i)[base4 + f];
            const float si = si_i[vv], q2 = q2_i[vv];
            pv_f4 g;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float inv = 1.f / (1.f - rho[c] * rho[c]);
                float t = w[r][c] / rj[c];                                                               // :296
                t -= 2.f * inv * rir[c] / (1.f + si);                                                    // :297
                t += inv * inv * ((1.f + rho[c] * rho[c]) * qij[c] - 2.f * rho[c] * si) / (1.f + q2);    // :298-299
                g[c] = t + hw[c];                                                                        // :300
            }
            if (ok) {
                reinterpret_cast<pv_f4*>(grad_o)[base4 + f] = g;
                if (grad2_o != nullptr) *reinterpret_cast<pv_f4*>(&grad2_o[v * (2 * Mp) + j0]) = g;
#pragma unroll
                for (int c = 0; c < 4; ++c) bj[c] += (double)(rho[c] * g[c]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // Bj partial of the block: the 64 / LPV lanes of a wave that hold the same factors, then the waves, in a fixed order
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int off = LPV; off < 64; off <<= 1) bj[c] += __shfl_xor(bj[c], off, 64);
    }
    if (lane < LPV) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bj_s[wave][j0 + c] = bj[c];
    }
    __syncthreads();
    if (tid < Mp) {
        double s = bj_s[0][tid];
        for (int k = 1; k < NW; ++k) s += bj_s[k][tid];
        bjpart[(int64_t)blockIdx.x * Mp + tid] = s;
    }
}

}  // namespace lcx
