// engine.hpp - the engine behind the C ABI declared in include/lcx.h, shared by its translation units.
//
// Owns the device state of one n_variables shard of a Linear CorEx fit (X, W and two moment sets, all resident in HBM) and
// enqueues the kernels of each dependency level of the reference's _calculate_moments_ns / _update_ns (linearcorex.py:236-334)
// on one HIP stream.  No torch types, no callbacks.  This header holds the context, the launch helpers and `Impl<T, CT>` - the
// typed implementation of every level, instantiated by whichever translation unit's entry points call it:
//     lcx_core.hip     handles, streams, exchange transport (RCCL / hook), first-contact self-test, state readback, timing
//     lcx_levels.hip   launch geometry, weights, the moment / update levels, lcx_iterate, the synergistic branch, moment readback
//                      (entry points; the typed work - most of the compile time - in lcx_levels_f32.hip / lcx_levels_f64.hip)
//     lcx_data.hip     upload + preprocess, the on-device generator, download, transform of new rows
//     lcx_outputs.hip  get_covariance, predict, invert
// (empirical.hip: gaussianize='empirical'.)  One object per unit, compiled side by side (__graft_entry__.py).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <sched.h>
#include <time.h>
#include <sys/mman.h>
#include <unistd.h>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is resolved at run time (see RcclApi)

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/lcx.h"
#include "gemm_kernels.hpp"
#include "gemm_split_kernels.hpp"
#include "moment_kernels.hpp"

using namespace lcx;

// empirical.hip: gaussianize='empirical' (:424-426) - per-column rank -> normal quantile, from a segmented sort of the
// transposed copy; rewrites both layouts of the shard
namespace lcx {
template <typename T> int empirical_columns(T* X, int64_t ldx, T* XT, int64_t Npad, int64_t N, int64_t V, hipStream_t st, std::string* err);
}

// -------------------------------------------------------------------------------------------------
// error plumbing
// -------------------------------------------------------------------------------------------------
inline thread_local std::string g_err;          // one per thread for the whole library (all translation units)
static int fail(int code, const std::string& msg) {
    g_err = msg;
    // HIP keeps the last error per thread until somebody reads it: a failed hipMalloc reported here would otherwise
    // resurface at the next kernel launch check (hipGetLastError) of a perfectly good call
    if (code == LCX_ERR_HIP) (void)hipGetLastError();
    return code;
}
#define HIPCHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(LCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + \
                                         __FILE__ + ":" + std::to_string(__LINE__) + ")");     \
    } while (0)
#define LCXCHECK(expr)              \
    do {                            \
        int s_ = (expr);            \
        if (s_ != LCX_OK) return s_; \
    } while (0)
#define KCHECK() HIPCHECK(hipGetLastError())

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// -------------------------------------------------------------------------------------------------
// exchange transport: who sums the exchange buffers over the ranks that share the variables axis
// -------------------------------------------------------------------------------------------------
// RCCL is bound at run time (dlopen of librccl.so.1, which is the copy a hosting process - PyTorch - has already loaded, so
// that both use one HIP runtime): a single-GPU user of liblcx_hip.so never needs it.
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool ok = false;
};
inline RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        for (const char* n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { api.error = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(lib, "ncclCommInitRank");
        api.AllReduce = (decltype(api.AllReduce))dlsym(lib, "ncclAllReduce");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(lib, "ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(lib, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.AllReduce && api.CommDestroy && api.GetErrorString;
        if (!api.ok) api.error = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
    });
    return api;
}
#define RCCLCHECK(expr)                                                                                         \
    do {                                                                                                        \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess)                                                                                  \
            return fail(LCX_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (" + __FILE__ + ":" + \
                                          std::to_string(__LINE__) + ")");                                      \
    } while (0)

// kind 0: none - the caller all-reduces the exchange buffers between the level calls (lcx_bind_exchange);
// kind 1: an RCCL communicator owned by the handle (lcx_comm_init): ncclAllReduce on the handle's stream;
// kind 2: the caller's function (lcx_set_exchange_hook): any other transport (MPI, gloo in the tests).
struct Transport {
    int kind = 0;
    ncclComm_t comm = nullptr;
    lcx_allreduce_fn hook = nullptr;
    void* user = nullptr;
    int rank = 0, nranks = 1;
};

// -------------------------------------------------------------------------------------------------
// context
// -------------------------------------------------------------------------------------------------
struct MomentSet {
    void *Y;                    // [Npad][Mp]  X.W^T (all-reduced), kept for the linear trial mode
    void *D;                    // [Vp][Mp]    X^T.Y of the shard
    void *rho, *rir, *qij;      // [Vp][Mp]
    void *si, *q2, *hscale;     // [Vp]
    double *uj, *ry, *wmag;     // small
    // synergistic branch (allocated on first use): X_i Z_j [Vp][Mp], X_i^2|Y [Vp], cy [Mp*Mp], Y_j^2 [Mp], 1/sd_j [Mp]
    void *xz, *x2y;
    double *cy, *yj2, *inv_sd;
    SetState* st;               // device
    SetState* hst;              // pinned host mirror (host address)
    SetState* hst_dev;          // same memory, device-visible address
    unsigned int seq_expect;    // sequence number of the last publication enqueued for this set
};

// scalar exchange buffer (doubles): [0] sum log(1+Si), [1] sum log(1+Qi-Si^2), [2] tangent partial, [3..8) spare,
// [SB_H, SB_H + Mp^2) H partial of the set last evaluated, [SB_H + Mp^2, + Mp + 8) detail sums (lcx_moments_detail /
// lcx_syn_moments_b).  One all-reduce of [0, SB_H + Mp^2) per moment evaluation carries everything the next update needs.
constexpr int SB_H = 8;
static inline int sb_det(int Mp) { return SB_H + Mp * Mp; }

// Staging of get_covariance (allocated on first use, kept for the life of the handle): two device row blocks and two
// pinned host blocks, so that the product of block k+1 overlaps the device-to-host copy of block k and the host-side
// placement of block k-1 into the caller's (pageable) matrix.
struct CovStage {
    void* dev[2] = {nullptr, nullptr};
    void* pin[2] = {nullptr, nullptr};
    size_t block_bytes = 0;
    int64_t block_rows = 0;
    void *op_a = nullptr, *op_b = nullptr, *std_dev = nullptr, *mean_dev = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_k[2] = {nullptr, nullptr}, ev_c[2] = {nullptr, nullptr}, t_a[2] = {nullptr, nullptr}, t_b[2] = {nullptr, nullptr};
    double last_kernel_seconds = 0.0;
};

struct TimingPair {
    hipEvent_t a, b;
    int kind;
};

struct lcx_ctx {
    int device, dtype;
    size_t es;
    int64_t N, V, Npad, ldx;    // ldx == Vp
    double Ndiv;                // the divisor of every sample moment (reference: self.n_samples, :249 / :260 / :355): N, unless the
                                // handle holds a batch that is evaluated with another fit's sample count (lcx_set_sample_divisor)
    int M, Mp, CT;
    hipStream_t own_stream, stream;
    void* X;                    // [Npad][ldx]
    void* XT;                   // [ldx][Npad] transposed copy (absent in single-copy mode)
    bool single_copy;           // X.B^T is read from the row-major X itself (gemm_cr): half the resident bytes, a 4-6 % slower pass
    bool panel;                 // X is the ONE panel-major copy [ldx / PW][Npad][PW] (gemm_kernels.hpp, PanelW): large shards whose two
                                // passes both run on the stream-K kernels; no transposed copy, both passes at full speed
    bool split;                 // float32 panel shards only: the two X passes run on the bf16 matrix pipe, every operand split exactly
                                // into three bf16 parts and 6 of the 9 partial products accumulated in float32 (gemm_split_kernels.hpp);
                                // off by default (LCX_F32_GEMM=split / lcx_set_f32_gemm)
    void* bsp;                  // the small operand of a pass in split form (split_b_kernel), 6 bytes per element
    size_t bsp_bytes;
    void* Wt[2];
    MomentSet set[2];
    void *grad, *update, *sgrad, *scratch;
    void *ydir, *ddir;          // Y(update) [Npad][Mp], D(update) [Vp][Mp]
    bool have_linear;
    void *ybuf_own, *ybuf;      // [Y (Npad Mp) | tail (Mp^2) | Y_g (Npad Mp), only with the merged pass]: ybuf_main = the first two
    int64_t ybuf_main;
    void* bjg;                  // [Mp] global Bj of the direction (merged pass under exchange: the tail is reused for W'.W'^T)
    double *sbuf_own, *sbuf;
    int64_t ybuf_elems, sbuf_elems;
    void *ypart, *dpart, *gpart, *gpartw;
    double *tcpart, *bjpart, *tanpart, *detpart, *ryinv, *invwork;
    SetState* states;           // device [2]
    SetState* host_states;      // pinned [2]
    int* order_dev;
    unsigned int* ticket;       // arrival counters: [0] small_moments_kernel, [1] moments_epilogue_kernel, [2] update_kernel
    bool full_sig;              // run the second pass of _sig (X^T.Y_g): the linear trial mode needs D(update)
    bool exchange;              // the exchange steps are live (several ranks, or forced for testing)
    bool w1_ready;              // Wt[1] already holds ws + update (written by update_kernel)
    // merged pass (float32 large shards with <= 64 padded factors): X.grad^T and the first trial's X.(ws+update)^T as ONE
    // pass over X with 2 Mp columns - half the X traffic of the two and the more efficient wide kernel
    bool merged_ok, y1_ready;   // y1_ready: ybuf / set[1].Y already hold Y of the eta = 1 trial
    // lcx_set_trial_reuse: the trials AFTER the first one of an iteration take X.w_update^T by linearity from what the iteration has
    // already computed exactly (Y of the current solution and X.update^T of lcx_update_c) instead of one more pass over X;
    // yk_ready: ybuf holds the Y of such a trial
    bool reuse_y, yk_ready;
    void *gw, *y2part, *ygbuf;  // [Vp][2 Mp] operand [grad | ws + update]; partial slots [S][Npad][2 Mp]; Y_g [Npad][Mp]
    int nt2_nb, nt2_nsuper, nt2_S;
    // launch geometry
    int nt_S, nt_KW, tn_S, tn_KW, gn_S, gv_S, pv_grid, target_waves, n_cus, nt_bpc, tn_bpc;
    int tan_blocks;             // per-block partials of update_tangent waiting in tanpart (summed by the next evaluation's tail)
    int tn_slots;               // partial slots of X^T.Y its consumers sum: tn_S, or 1 when tn_big pre-reduces many slots
    // column-tiled stream-K kernel (gemm_ct) per pass: used when the shard has enough column tiles
    bool nt_ct, tn_ct;
    bool f64_4x4;               // float64, n_hidden <= 32: the small-shard passes run on v_mfma_f64_4x4x4 (gemm_tn4)
    int nt_nb, nt_nsuper, tn_nb, tn_nsuper;
    // timing
    bool timing;
    int t_every, t_count;       // HIP-event timing samples every t_every-th X pass (an event pair costs ~5 us of stream time)
    std::vector<TimingPair> pending;
    std::vector<TimingPair> pool;
    int64_t t_launch[3];        // kind 0 = X.B^T, 1 = X^T.Y, 2 = the merged X.[grad | ws+update]^T pass (2 Mp columns)
    int64_t t_pass[3];          // every X pass issued while timing is on (sampled or not)
    double t_ms[3];
    double t_max[3];            // longest timed launch per kind; launches below a fifth of it were skipped by their flag
    int64_t t_skipped[3];
    bool have_direction;
    int world;                  // ranks sharing the variables axis (1: no exchange between levels)
    unsigned int seq_next;
    CovStage* cov;
    // lcx_iterate: the direction and first trial of the NEXT iteration are already enqueued (speculation); spec_dirty: a
    // speculation was abandoned, i.e. sbuf holds the H of a trial that was never accepted instead of the H of set 0
    bool spec_pending, spec_dirty;
    // lcx_iterate computes the gradient of every trial right behind its evaluation (early_grad: `grad` / `bjpart` hold the
    // gradient of set 1); when the trial is accepted it IS the next iteration's gradient (grad_ready) and the GPU did not
    // wait for the host's decision to start on it
    bool early_grad, grad_ready;
    double spec_eps;
    size_t bytes_resident;      // device bytes owned by the handle (X, its transposed copy, moments, work space)
    // LCX_Y_PIPELINE=chunks[:n]: the Y all-reduce of lcx_moments_a in n row chunks on a second stream, each behind the event of its
    // chunk's slot reduction (and, for the wave-split kernels, behind its own row chunk of the pass); default off
    int ypipe;
    bool ypipe_force_pass;
    hipStream_t comm_stream;
    hipEvent_t ypipe_ev[17];
    Transport tr;               // in-library exchange (kind != 0): every level sums what it produced over the ranks itself
    int64_t n_exchanges;        // all-reduces issued by the library (diagnostics)
};

template <typename T> static inline T* P(void* p) { return reinterpret_cast<T*>(p); }
static int wait_published(lcx_ctx* h, MomentSet& s);
static inline void cancel_speculation(lcx_ctx* h);

// Sum `count` elements at `buf` (device memory) over the ranks, in place, stream-ordered with the handle's kernels.
// Without a bound transport this is the caller's job between the level calls; without exchange steps there is nothing to sum.
static int exchange_on(lcx_ctx* h, hipStream_t st, void* buf, int64_t count, int dtype) {
    if (!h->exchange || h->tr.kind == 0 || count <= 0) return LCX_OK;
    h->n_exchanges += 1;
    if (h->tr.kind == 1) {
        RCCLCHECK(rccl().AllReduce(buf, buf, (size_t)count, dtype == LCX_F32 ? ncclFloat : ncclDouble, ncclSum, h->tr.comm, st));
        return LCX_OK;
    }
    const int rc = h->tr.hook(h->tr.user, buf, count, dtype, (void*)st);
    if (rc != 0) return fail(LCX_ERR_COMM, "the exchange hook reported failure " + std::to_string(rc));
    return LCX_OK;
}
static int exchange(lcx_ctx* h, void* buf, int64_t count, int dtype) { return exchange_on(h, h->stream, buf, count, dtype); }
// the library can sequence whole iterations when it does not depend on the caller for the sums
static inline bool self_contained(const lcx_ctx* h) { return !h->exchange || h->tr.kind != 0 || h->world == 1; }

// ---- first contact with a transport: does it sum, and does every rank get the same bits? --------------------------
// pattern 0: small integers (exact in both precisions, the sum over the ranks is known in closed form);
// pattern 1: rank-dependent pseudo-random values over 12 binades (the sum depends on the order of the reduction: what is
//            checked is that every rank ends up with the SAME bits, which the rank-identical line-search decisions need)
namespace lcx {
__device__ __forceinline__ uint64_t st_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ void selftest_fill_kernel(T* __restrict__ buf, int64_t n, int pattern, int rank) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (pattern == 0) {
            buf[i] = (T)((double)((i % 251) + 1) * (double)(rank + 1));
        } else {
            const uint64_t hsh = st_mix((uint64_t)i * 0xD1342543DE82EF95ull + (uint64_t)rank);
            const double mant = (double)(hsh >> 11) * (1.0 / 9007199254740992.0) - 0.5;       // [-0.5, 0.5)
            buf[i] = (T)ldexp(mant, (int)((hsh & 1023) % 12) - 6);
        }
    }
}
// out[0] = number of elements that differ from the closed-form sum (pattern 0 only), out[1] = order-independent hash of the bits
template <typename T>
__global__ void selftest_check_kernel(const T* __restrict__ buf, int64_t n, int pattern, int nranks, unsigned long long* __restrict__ out) {
    unsigned long long bad = 0, hsh = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T v = buf[i];
        if (pattern == 0) {
            const T want = (T)((double)((i % 251) + 1) * (double)(nranks * (nranks + 1) / 2));
            bad += (v != want) ? 1ull : 0ull;
        }
        uint64_t bits = 0;
        if constexpr (sizeof(T) == 4) bits = (uint64_t)__float_as_uint((float)v); else bits = (uint64_t)__double_as_longlong((double)v);
        hsh += st_mix(bits ^ st_mix((uint64_t)i));
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], hsh);
}
}  // namespace lcx

// Temporary device buffers of one call: freed on every return path (an OOM in the middle of a call must not leak the
// buffers allocated before it - that is exactly when memory matters).
struct DevTemps {
    std::vector<void*> ptrs;
    ~DevTemps() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename U> int get(U** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) {
            *out = nullptr;
            return fail(LCX_ERR_HIP, std::string("hipMalloc of a temporary of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
        }
        ptrs.push_back(p);
        *out = reinterpret_cast<U*>(p);
        return LCX_OK;
    }
};

// -------------------------------------------------------------------------------------------------
// GEMM launchers
// -------------------------------------------------------------------------------------------------
static int timing_begin(lcx_ctx* h, int kind, TimingPair* tp) {
    tp->kind = -1;
    if (!h || !h->timing || kind < 0) return LCX_OK;
    h->t_pass[kind] += 1;
    if (h->t_every > 1 && (h->t_count++ % h->t_every) != 0) return LCX_OK;
    if (h->pool.empty()) {
        HIPCHECK(hipEventCreate(&tp->a));
        HIPCHECK(hipEventCreate(&tp->b));
    } else {
        *tp = h->pool.back();
        h->pool.pop_back();
    }
    tp->kind = kind;
    HIPCHECK(hipEventRecord(tp->a, h->stream));
    return LCX_OK;
}
static int timing_end(lcx_ctx* h, int kind, TimingPair* tp) {
    if (!h || !h->timing || kind < 0 || tp->kind < 0) return LCX_OK;
    HIPCHECK(hipEventRecord(tp->b, h->stream));
    h->pending.push_back(*tp);
    return LCX_OK;
}
static int timing_collect(lcx_ctx* h) {
    // The X^T.Y pass of an invalid trial (:250-251) returns at its first instruction (skip flag): such a launch is not a pass
    // and must not pull the average down.  A launch shorter than a fifth of the longest one of its kind is counted apart.
    std::vector<float> dur(h->pending.size());
    for (size_t k = 0; k < h->pending.size(); ++k) {
        TimingPair& tp = h->pending[k];
        HIPCHECK(hipEventSynchronize(tp.b));
        HIPCHECK(hipEventElapsedTime(&dur[k], tp.a, tp.b));
        if (dur[k] > h->t_max[tp.kind]) h->t_max[tp.kind] = dur[k];
    }
    for (size_t k = 0; k < h->pending.size(); ++k) {
        TimingPair& tp = h->pending[k];
        if (dur[k] < 0.2 * h->t_max[tp.kind]) {
            h->t_skipped[tp.kind] += 1;
        } else {
            h->t_launch[tp.kind] += 1;
            h->t_ms[tp.kind] += dur[k];
        }
        h->pool.push_back(tp);
    }
    h->pending.clear();
    return LCX_OK;
}

// Dynamic LDS above 48 KiB needs an explicit opt-in per kernel - once per (device, function, size), not per launch: the
// attribute call is a driver round trip on the path between a trial's result and the next launch.
template <typename F> static int allow_lds(F* f, size_t bytes) {
    if (bytes <= 48 * 1024) return LCX_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    const std::pair<int, const void*> key(dev, (const void*)f);
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return LCX_OK;
    HIPCHECK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[key] = bytes;
    return LCX_OK;
}

template <typename T, int CT>
static int launch_nt(hipStream_t st, const T* X, int64_t ldx, int64_t rows_pad, const T* B, T* out,
                     int S, int KW, const int* skip) {
    constexpr int RT = NtShape<T, CT>::RT;
    constexpr int CH = 4 * (32 / (int)sizeof(T));
    const int nchunks = (int)(ldx / CH);
    dim3 grid((unsigned)(rows_pad / (16 * RT)), (unsigned)S);
    const size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    switch (KW) {
        case 1: hipLaunchKernelGGL((gemm_nt_kernel<T, CT, RT, 1>), grid, dim3(64), lds, st, X, ldx, B, out, rows_pad, nchunks, S, skip); break;
        case 2: hipLaunchKernelGGL((gemm_nt_kernel<T, CT, RT, 2>), grid, dim3(128), lds, st, X, ldx, B, out, rows_pad, nchunks, S, skip); break;
        case 4: hipLaunchKernelGGL((gemm_nt_kernel<T, CT, RT, 4>), grid, dim3(256), lds, st, X, ldx, B, out, rows_pad, nchunks, S, skip); break;
        default: hipLaunchKernelGGL((gemm_nt_kernel<T, CT, RT, 8>), grid, dim3(512), lds, st, X, ldx, B, out, rows_pad, nchunks, S, skip); break;
    }
    KCHECK();
    return LCX_OK;
}

// out rows (padded "v" count) must be a multiple of 16*RT; K a multiple of 16.
// 8-wave blocks do not exist for 256 padded factors (512 threads leave 256 registers per lane for a 128-register tile
// plus its operand buffers): those shapes use at most 4 waves
template <int CT> struct MaxKw { static constexpr int v = CT >= 16 ? 4 : 8; };

template <typename T, int CT, int RT, bool SCALE, bool NTA = false>
static int launch_tn(hipStream_t st, const T* A, int64_t lda, int64_t K, int64_t vcols_pad, const T* B,
                     const T* rowscale, T* out, int S, int KW, const int* skip, int64_t out_rows = 0) {
    // out_rows: rows of one partial slot of `out` when this launch covers only `vcols_pad` of them (a row chunk of the pass)
    if (out_rows <= 0) out_rows = vcols_pad;
    const int kgroups = (int)(K / 16);
    dim3 grid((unsigned)(vcols_pad / (16 * RT)), (unsigned)S);
    if (KW > 4 && CT >= 16) KW = 4;          // 8 partial tiles of 256 factors do not fit the LDS
    const size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    if (lds > 48 * 1024) {
        switch (KW) {
            case 1: LCXCHECK(allow_lds(gemm_tn_kernel<T, CT, RT, 1, SCALE, 4, NTA>, lds)); break;
            case 2: LCXCHECK(allow_lds(gemm_tn_kernel<T, CT, RT, 2, SCALE, 4, NTA>, lds)); break;
            case 4: LCXCHECK(allow_lds(gemm_tn_kernel<T, CT, RT, 4, SCALE, 4, NTA>, lds)); break;
            default: LCXCHECK(allow_lds(gemm_tn_kernel<T, CT, RT, MaxKw<CT>::v, SCALE, 4, NTA>, lds)); break;
        }
    }
    switch (KW) {
        case 1: hipLaunchKernelGGL((gemm_tn_kernel<T, CT, RT, 1, SCALE, 4, NTA>), grid, dim3(64), lds, st, A, lda, (int64_t)(16 * RT), B, rowscale, out, out_rows, kgroups, S, skip); break;
        case 2: hipLaunchKernelGGL((gemm_tn_kernel<T, CT, RT, 2, SCALE, 4, NTA>), grid, dim3(128), lds, st, A, lda, (int64_t)(16 * RT), B, rowscale, out, out_rows, kgroups, S, skip); break;
        case 4: hipLaunchKernelGGL((gemm_tn_kernel<T, CT, RT, 4, SCALE, 4, NTA>), grid, dim3(256), lds, st, A, lda, (int64_t)(16 * RT), B, rowscale, out, out_rows, kgroups, S, skip); break;
        default: hipLaunchKernelGGL((gemm_tn_kernel<T, CT, RT, MaxKw<CT>::v, SCALE, 4, NTA>), grid, dim3(64 * MaxKw<CT>::v), lds, st, A, lda, (int64_t)(16 * RT), B, rowscale, out, out_rows, kgroups, S, skip); break;
    }
    KCHECK();
    return LCX_OK;
}

// gemm_tn4 (float64 on v_mfma_f64_4x4x4): same grid / partial-tile contract as launch_tn; KW in {2, 4}
template <int CT>
static int launch_tn4(hipStream_t st, const double* A, int64_t lda, int64_t K, int64_t vcols_pad, const double* B, double* out, int S,
                      int KW, const int* skip, int64_t out_rows = 0) {
    if (out_rows <= 0) out_rows = vcols_pad;
    constexpr int RT = TnShape<double, CT>::RT, U = 4;
    dim3 grid((unsigned)(vcols_pad / (16 * RT)), (unsigned)S);
    const int kgroups = (int)(K / 16);
    if (KW == 2) {
        const size_t lds = Tn4Lds<CT, RT, 2, U>::bytes;
        LCXCHECK(allow_lds(gemm_tn4_kernel<CT, RT, 2, U, true>, lds));
        hipLaunchKernelGGL((gemm_tn4_kernel<CT, RT, 2, U, true>), grid, dim3(128), lds, st, A, lda, B, out, out_rows, kgroups, S, skip);
    } else {
        const size_t lds = Tn4Lds<CT, RT, 4, U>::bytes;
        LCXCHECK(allow_lds(gemm_tn4_kernel<CT, RT, 4, U, true>, lds));
        hipLaunchKernelGGL((gemm_tn4_kernel<CT, RT, 4, U, true>), grid, dim3(256), lds, st, A, lda, B, out, out_rows, kgroups, S, skip);
    }
    KCHECK();
    return LCX_OK;
}

// The same pass on the bf16 matrix pipe (float32 panel shards in split mode): B is split once (split_b_kernel), then the contraction
// kernel with the slot contract of launch_ct / launch_cr (out[slot][rows][Mp], `maxslots` slots, unused ones zero-filled) on a
// geometry of its own: 64 and 128 columns run 8-wave blocks of 512 rows, one per CU (tools/gemm_probe9: -3 % against 2 x 4 waves at
// 64 columns - B is re-read once per super tile, and every block must own at least one unit), so nsuper and nb are re-derived here;
// the slot count cdiv(nb, nsuper) + 1 then stays within what lcx_create allocated for the float32 kernels (half the blocks on half
// the super tiles).  Two 32-element steps per barrier where the LDS image allows it (up to 64 columns); MFMA phase at raised wave
// priority (-1 ... -6 %).
template <int CT> struct SplitShape {
    static constexpr int KW = CT >= 4 ? 8 : 4;
    static constexpr int KS = CT <= 4 ? 2 : 1;
    static constexpr int PRIO = CT == 8 ? 2 : 1;
    static constexpr int BPC = KW == 8 ? 1 : 2;                  // resident blocks per CU (registers: 2 waves per SIMD)
};
template <int CT, bool CONTRACT_N>
static int launch_split(hipStream_t st, const float* A, int64_t ps, int64_t K, int64_t rows, const float* B, float* out, int nb_f32, int maxslots,
                        const int* skip, void* bsp, int n_cus) {
    if constexpr (CT == 2 || CT == 4 || CT == 8) {
        constexpr int KW = SplitShape<CT>::KW, KS = SplitShape<CT>::KS;
        const int ng32 = (int)(K / SPLIT_KG), ng = ng32 / KS;          // K is a multiple of 64 (padded sizes)
        u32x4_t* sp = reinterpret_cast<u32x4_t*>(bsp);
        const int64_t tasks = (int64_t)ng32 * 64 * CT;
        hipLaunchKernelGGL((split_b_kernel<CT, CONTRACT_N>), dim3((unsigned)(cdiv(tasks, 256) < 2048 ? cdiv(tasks, 256) : 2048)), dim3(256), 0, st, B, sp,
                           ng32, skip);
        const int nsuper = (int)cdiv(rows, KW * 64);
        const int64_t total = (int64_t)nsuper * ng;
        int64_t nb = (int64_t)n_cus * SplitShape<CT>::BPC;
        if (nb > nb_f32) nb = nb_f32;                                  // a forced block count (LCX_CT_NB) binds this launch too
        if (nb > (int64_t)(maxslots - 1) * nsuper) nb = (int64_t)(maxslots - 1) * nsuper;
        if (nb > total) nb = total;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL((gemm_split_kernel<CT, KW, 6, CONTRACT_N, true, false, 2, KS, SplitShape<CT>::PRIO>), dim3((unsigned)nb), dim3(64 * KW), 0, st,
                           A, ps, (const u32x4_t*)sp, out, rows, rows, ng, nsuper, maxslots, skip);
        KCHECK();
        return LCX_OK;
    } else {
        return fail(LCX_ERR_ARG, "launch_split: unsupported factor count");
    }
}
// split mode exists for float32 with 32 / 64 / 128 padded factors (and the merged pass of the first two)
template <typename T, int CT> static constexpr bool split_capable() { return sizeof(T) == 4 && (CT == 2 || CT == 4 || CT == 8); }
// LCX_PV_MFMA=0 keeps the thread-per-(variable, factor) forms of moments_epilogue / grad where the matrix-pipe forms exist
// (read at every launch - two per iteration - so that a test can compare both forms inside one process)
static inline bool pv_mfma() {
    const char* e = getenv("LCX_PV_MFMA");
    return !(e && *e && atoi(e) == 0);
}
// waves per block of the stream-K kernels: CtShape's; at 128 float32 columns LCX_CT8_KW=4|8 forces either for A/B runs.  8 waves on one
// copy of B pay on BOTH layouts there (tools/layout_ab.sh, profiles/r05_layout_ab_one_box.txt: config-4 shard row-major 19.6-19.7 ->
// 20.1-20.2 it/s, panel-major 19.5-19.6 -> 20.2; the "8 waves lose 10-20 % on the row-major layouts" of an earlier kernel generation
// no longer holds), so the layout does not enter; `panel` only names the instantiation whose occupancy sizes the grid (ct_geometry)
template <typename T, int CT> static inline int ct_kw(bool panel) {
    (void)panel;
    if constexpr (sizeof(T) == 4 && CT == 8) {
        static const int forced = []() {
            const char* e = getenv("LCX_CT8_KW");
            return (e && *e) ? (atoi(e) == 8 ? 8 : 4) : 0;
        }();
        if (forced) return forced;
    }
    return CtShape<T, CT>::KW;
}
// gemm_ct launch: nb balanced blocks over (super tile, group) units; partial tiles -> out[slot][out_rows][Mp]
template <typename T, int CT, bool PANEL = false>
static int launch_ct(hipStream_t st, const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int nb,
                     int nsuper, int maxslots, const int* skip, void* bsp = nullptr, int n_cus = 0) {
    typedef CtShape<T, CT> S;
    if constexpr (PANEL && split_capable<T, CT>()) {
        if (bsp) return launch_split<CT, true>(st, A, lda, K, vcols, B, out, nb, maxslots, skip, bsp, n_cus);
    }
    const int ng = (int)(K / (4 * S::U));
    if constexpr (sizeof(T) == 4 && CT == 8) {
        if (ct_kw<T, CT>(PANEL) == 8)
            hipLaunchKernelGGL((gemm_ct_kernel<T, CT, S::RT, 8, S::U, true, PANEL>), dim3((unsigned)nb), dim3(512), 0, st, A, lda, B, out, vcols, vcols,
                               ng, nsuper, maxslots, skip);
        else
            hipLaunchKernelGGL((gemm_ct_kernel<T, CT, S::RT, 4, S::U, true, PANEL>), dim3((unsigned)nb), dim3(256), 0, st, A, lda, B, out, vcols, vcols,
                               ng, nsuper, maxslots, skip);
    } else {
        hipLaunchKernelGGL((gemm_ct_kernel<T, CT, S::RT, S::KW, S::U, true, PANEL>), dim3((unsigned)nb), dim3(64 * S::KW), 0, st, A, lda, B, out,
                           vcols, vcols, ng, nsuper, maxslots, skip);
    }
    KCHECK();
    return LCX_OK;
}
// gemm_cr launch (X.B^T from the row-major shard itself): same unit / slot contract as launch_ct
// (PANEL: the panel-major copy, lda = the panel stride, non-temporal loads - every line is read once per pass)
template <typename T, int CT, bool PANEL = false>
static int launch_cr(hipStream_t st, const T* A, int64_t lda, int64_t K, int64_t nrows, const T* B, T* out, int nb, int nsuper,
                     int maxslots, const int* skip, void* bsp = nullptr, int n_cus = 0) {
    typedef CtShape<T, CT> S;
    if constexpr (PANEL && split_capable<T, CT>()) {
        if (bsp) return launch_split<CT, false>(st, A, lda, K, nrows, B, out, nb, maxslots, skip, bsp, n_cus);
    }
    const int ng = (int)(K / (4 * S::U));
    if constexpr (sizeof(T) == 4 && CT == 8) {
        if (ct_kw<T, CT>(PANEL) == 8)
            hipLaunchKernelGGL((gemm_cr_kernel<T, CT, S::RT, 8, S::U, PANEL, PANEL>), dim3((unsigned)nb), dim3(512), 0, st, A, lda, B, out, nrows, nrows,
                               ng, nsuper, maxslots, skip);
        else
            hipLaunchKernelGGL((gemm_cr_kernel<T, CT, S::RT, 4, S::U, PANEL, PANEL>), dim3((unsigned)nb), dim3(256), 0, st, A, lda, B, out, nrows, nrows,
                               ng, nsuper, maxslots, skip);
    } else {
        hipLaunchKernelGGL((gemm_cr_kernel<T, CT, S::RT, S::KW, S::U, PANEL, PANEL>), dim3((unsigned)nb), dim3(64 * S::KW), 0, st, A, lda, B, out, nrows,
                           nrows, ng, nsuper, maxslots, skip);
    }
    KCHECK();
    return LCX_OK;
}
// geometry of a stream-K launch over `vcols` output rows and K contraction elements.  panel / cr name the instantiation that will be
// launched (gemm_ct or gemm_cr, on the panel-major copy or a row-major one): its wave count and ITS occupancy size the grid
template <typename T, int CT, int KW, bool PANEL, bool CR> static hipError_t ct_occupancy(int* bpc) {
    typedef CtShape<T, CT> S;
    if constexpr (CR) return hipOccupancyMaxActiveBlocksPerMultiprocessor(bpc, (const void*)gemm_cr_kernel<T, CT, S::RT, KW, S::U, PANEL, PANEL>, 64 * KW, 0);
    else return hipOccupancyMaxActiveBlocksPerMultiprocessor(bpc, (const void*)gemm_ct_kernel<T, CT, S::RT, KW, S::U, true, PANEL>, 64 * KW, 0);
}
template <typename T, int CT>
static void ct_geometry(int n_cus, int64_t K, int64_t vcols, int force_nb, int* nb_o, int* nsuper_o, int* slots_o, bool panel = false,
                        bool cr = false) {
    typedef CtShape<T, CT> S;
    const int KW = ct_kw<T, CT>(panel);
    int bpc = 0;
    hipError_t oe;
    if constexpr (sizeof(T) == 4 && CT == 8) {
        if (KW == 8) oe = panel ? (cr ? ct_occupancy<T, CT, 8, true, true>(&bpc) : ct_occupancy<T, CT, 8, true, false>(&bpc))
                                : (cr ? ct_occupancy<T, CT, 8, false, true>(&bpc) : ct_occupancy<T, CT, 8, false, false>(&bpc));
        else oe = panel ? (cr ? ct_occupancy<T, CT, 4, true, true>(&bpc) : ct_occupancy<T, CT, 4, true, false>(&bpc))
                        : (cr ? ct_occupancy<T, CT, 4, false, true>(&bpc) : ct_occupancy<T, CT, 4, false, false>(&bpc));
    } else {
        oe = panel ? (cr ? ct_occupancy<T, CT, S::KW, true, true>(&bpc) : ct_occupancy<T, CT, S::KW, true, false>(&bpc))
                   : (cr ? ct_occupancy<T, CT, S::KW, false, true>(&bpc) : ct_occupancy<T, CT, S::KW, false, false>(&bpc));
    }
    if (oe != hipSuccess || bpc < 1) bpc = 1;
    // measured (tools/gemm_probe4, 50k x 20k float32, n_hidden 64): 2 resident blocks per CU 122 TF/s, 3 blocks 118
    {
        const char* e = getenv("LCX_CT_BPC");
        const int cap = (e && *e) ? atoi(e) : 2;
        if (cap > 0 && bpc > cap) bpc = cap;
    }
    const int nsuper = (int)cdiv(vcols, KW * 16 * S::RT);
    const int64_t total = (int64_t)nsuper * (K / (4 * S::U));
    int64_t nb = force_nb > 0 ? force_nb : (int64_t)n_cus * bpc;
    if (nb > total) nb = total;
    if (nb < 1) nb = 1;
    *nb_o = (int)nb;
    *nsuper_o = nsuper;
    *slots_o = (int)cdiv(nb, nsuper) + 1;
}

static int pick_split(int64_t tiles, int kw, int64_t kunits, int target_waves, int cap) {
    int64_t s = cdiv(target_waves, tiles * kw);
    const int64_t by_work = kunits / ((int64_t)kw * 4);   // keep >= 4 contraction units per wave
    if (s > by_work) s = by_work;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}
static int pick_kw(int64_t kunits) { return kunits >= 32 ? 4 : (kunits >= 8 ? 2 : 1); }

template <typename T, int CT> struct Geo {
    static constexpr int NT_RT = NtShape<T, CT>::RT;
    static constexpr int TN_RT = TnShape<T, CT>::RT;
    // Gram matrices contract a [K][Mp] array with itself: the column tile must fit in Mp
    static constexpr int G_RT = TnShape<T, CT>::RT < CT ? TnShape<T, CT>::RT : CT;
    static constexpr int CH = 4 * (32 / (int)sizeof(T));
};


// -------------------------------------------------------------------------------------------------
// typed implementation
// -------------------------------------------------------------------------------------------------
template <typename T, int CT> struct Impl {
    static constexpr int Mp = 16 * CT;
    // more than 256 padded factors: the untuned wide path - every contraction on gemm_wide (run-time strides, the factor axis
    // tiled like any other), the per-variable kernels with one thread per factor
    static constexpr bool WIDE = CT > 16;
    static constexpr int NTV = Pvt<Mp>::v;          // threads per block of the per-variable kernels
    static constexpr int VPB = NTV / Mp;
    static constexpr int DT = sizeof(T) == 4 ? LCX_F32 : LCX_F64;

    // C[z][M][ldc] = sum over split z of opA . B (. rowscale) on gemm_wide; M, N multiples of 64, K of 16
    template <bool TRANS_A, bool SCALE>
    static int wide_gemm(lcx_ctx* h, const T* A, int64_t lda, const T* B, int64_t ldb, const T* scale, T* C, int64_t ldc, int64_t M,
                         int64_t N, int64_t K, int S, const int* skip) {
        dim3 grid((unsigned)(N / 64), (unsigned)(M / 64), (unsigned)S);
        hipLaunchKernelGGL((gemm_wide_kernel<T, TRANS_A, SCALE>), grid, dim3(256), 0, h->stream, A, lda, B, ldb, scale, C, ldc, M, K, S, skip);
        KCHECK();
        return LCX_OK;
    }

    // resident blocks per CU of a kernel at a given block size / dynamic LDS
    template <typename F> static int blocks_per_cu(F* f, int threads, size_t lds) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)f, threads, lds) != hipSuccess || n < 1) n = 1;
        return n;
    }
    static int env_int(const char* name, int dflt) {
        const char* v = getenv(name);
        return (v && *v) ? atoi(v) : dflt;
    }
    // Split of the contraction.  These launches are HBM-bound and every block lives for the whole
    // launch, so what matters is how full the last "round" of resident blocks is: 1.03 rounds cost
    // almost 2 (measured: 632 blocks on 512 slots 102 us vs 474 blocks 79 us).  Pick the split that
    // fills whole rounds best, with a small penalty per extra split (partial tiles to write + sum).
    static int single_round_split(int64_t tiles, int64_t capacity_blocks, int64_t kunits, int kw, int cap) {
        if (tiles < 1) tiles = 1;
        if (capacity_blocks < 1) capacity_blocks = 1;
        const int64_t work_cap = kunits / ((int64_t)kw * 4);           // keep >= 4 contraction units per wave
        int64_t by_work = work_cap;
        if (by_work > cap) by_work = cap;
        if (by_work < 1) by_work = 1;
        int best = 1;
        double best_score = -1.0;
        for (int s = 1; s <= by_work; ++s) {
            const int64_t blocks = tiles * s;
            const int64_t rounds = (blocks + capacity_blocks - 1) / capacity_blocks;
            const double score = (double)blocks / (double)(rounds * capacity_blocks) - 0.015 * s;
            if (score > best_score + 1e-9) { best_score = score; best = s; }
        }
        if (tiles * best * 2 >= capacity_blocks) return best;
        // Few tiles (n_samples << n_variables for X.W^T, or the reverse for X^T.Y): the rule above leaves most of
        // the chip idle (measured: 448 x 20000, 7 tiles, 1 split: 334 us for a 72 MB pass).  Time of the pass in
        // units of a full-chip stream = max(1, rounds * capacity / blocks), plus what the partial tiles cost to write
        // and sum back: per split 2 * Mp / K of the X bytes and a fixed term.
        int64_t hi = work_cap < 64 ? work_cap : 64;                     // consumers sum the slots serially per element
        if (hi < 1) hi = 1;
        double best_cost = 1e30;
        const double per_split = 2.0 * Mp / ((double)kunits * 16.0) + 0.004;
        for (int s = 1; s <= hi; ++s) {
            const int64_t blocks = tiles * s;
            const int64_t rounds = (blocks + capacity_blocks - 1) / capacity_blocks;
            double t = (double)(rounds * capacity_blocks) / (double)blocks;
            if (t < 1.0) t = 1.0;
            const double cost = t + per_split * s;
            if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
        }
        return best;
    }

    static int geometry(lcx_ctx* h) {
        if constexpr (WIDE) {
            h->nt_S = h->tn_S = h->tn_slots = 1;
            h->nt_KW = h->tn_KW = 4;
            h->nt_bpc = h->tn_bpc = 1;
            h->nt_ct = h->tn_ct = h->f64_4x4 = h->merged_ok = h->panel = false;
            h->nt_nb = h->nt_nsuper = h->tn_nb = h->tn_nsuper = h->nt2_nb = h->nt2_nsuper = h->nt2_S = 0;
            // Gram matrices: (Mp / 64)^2 tiles; split the contraction until the chip is about twice covered
            auto gsplit = [&](int64_t K) {
                const int64_t tiles = (int64_t)(Mp / 64) * (Mp / 64);
                int64_t sp = cdiv(2 * (int64_t)h->n_cus, tiles);
                if (sp > K / 64) sp = K / 64;
                if (sp > 64) sp = 64;
                return (int)(sp < 1 ? 1 : sp);
            };
            h->gn_S = gsplit(h->Npad);
            h->gv_S = gsplit(h->ldx);
            h->pv_grid = (int)(h->V < 1024 ? h->V : 1024);
            return LCX_OK;
        } else {
            return geometry_tuned(h);
        }
    }
    static int geometry_tuned(lcx_ctx* h) {
        constexpr int NT_RT = Geo<T, CT>::NT_RT, TN_RT = Geo<T, CT>::TN_RT;
        const int64_t nchunks = h->ldx / Geo<T, CT>::CH;
        const int64_t kgn = h->Npad / 16, kgv = h->ldx / 16;
        const int cus = h->n_cus;
        // X . B^T, computed as XT^T . B with the tn kernel: tiles over n, contraction over v
        h->nt_KW = env_int("LCX_NT_KW", pick_kw(kgv));
        // float64 with <= 32 factors: v_mfma_f64_4x4x4 (72 TF/s measured) instead of 16x16x4 (47.6 TF/s)
        h->f64_4x4 = sizeof(T) == 8 && CT <= 2 && env_int("LCX_F64_MFMA", 4) == 4;
        if (h->f64_4x4) {
            if constexpr (sizeof(T) == 8 && CT <= 2) {
                if (h->nt_KW != 2) h->nt_KW = 4;
                const int bpc = h->nt_KW == 2 ? blocks_per_cu(gemm_tn4_kernel<CT, TN_RT, 2, 4, true>, 128, Tn4Lds<CT, TN_RT, 2, 4>::bytes)
                                              : blocks_per_cu(gemm_tn4_kernel<CT, TN_RT, 4, 4, true>, 256, Tn4Lds<CT, TN_RT, 4, 4>::bytes);
                h->nt_bpc = bpc;
                h->nt_S = env_int("LCX_NT_S", single_round_split(h->Npad / (16 * TN_RT), (int64_t)bpc * cus, kgv, h->nt_KW, 16));
            }
        } else {
            const size_t lds = (size_t)h->nt_KW * 16 * TN_RT * Mp * sizeof(T);
            int bpc = 1;
            switch (h->nt_KW) {
                case 1: bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, 1, false>, 64, lds); break;
                case 2: bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, 2, false>, 128, lds); break;
                case 4: bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, 4, false>, 256, lds); break;
                default: h->nt_KW = MaxKw<CT>::v; bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, MaxKw<CT>::v, false>, 64 * MaxKw<CT>::v, (size_t)MaxKw<CT>::v * 16 * TN_RT * Mp * sizeof(T)); break;
            }
            h->nt_bpc = bpc;
            h->nt_S = env_int("LCX_NT_S", single_round_split(h->Npad / (16 * TN_RT), (int64_t)bpc * cus, kgv, h->nt_KW, 16));
        }
        // X^T . Y
        h->tn_KW = env_int("LCX_TN_KW", pick_kw(kgn));
        if (h->f64_4x4) {
            if constexpr (sizeof(T) == 8 && CT <= 2) {
                if (h->tn_KW != 2) h->tn_KW = 4;
                const int bpc = h->tn_KW == 2 ? blocks_per_cu(gemm_tn4_kernel<CT, TN_RT, 2, 4, true>, 128, Tn4Lds<CT, TN_RT, 2, 4>::bytes)
                                              : blocks_per_cu(gemm_tn4_kernel<CT, TN_RT, 4, 4, true>, 256, Tn4Lds<CT, TN_RT, 4, 4>::bytes);
                h->tn_bpc = bpc;
                h->tn_S = env_int("LCX_TN_S", single_round_split(h->ldx / (16 * TN_RT), (int64_t)bpc * cus, kgn, h->tn_KW, 32));
            }
        } else {
            const size_t lds = (size_t)h->tn_KW * 16 * TN_RT * Mp * sizeof(T);
            int bpc = 1;
            switch (h->tn_KW) {
                case 1: bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, 1, false>, 64, lds); break;
                case 2: bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, 2, false>, 128, lds); break;
                case 4: bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, 4, false>, 256, lds); break;
                default: h->tn_KW = MaxKw<CT>::v; bpc = blocks_per_cu(gemm_tn_kernel<T, CT, TN_RT, MaxKw<CT>::v, false>, 64 * MaxKw<CT>::v, (size_t)MaxKw<CT>::v * 16 * TN_RT * Mp * sizeof(T)); break;
            }
            h->tn_bpc = bpc;
            h->tn_S = env_int("LCX_TN_S", single_round_split(h->ldx / (16 * TN_RT), (int64_t)bpc * cus, kgn, h->tn_KW, 32));
        }
        // Large shards: column-tiled stream-K kernel with B staged through LDS (gemm_ct).  It writes
        // ceil(blocks / super tiles) + 1 partial slots, so it only pays when there are many column tiles.
        {
            const char* force = getenv("LCX_GEMM");          // "ct" / "tn" force one kernel for both passes
            // When does gemm_ct (B shared through LDS, stream-K) beat the wave-split kernel although it writes more
            // partial slots?  From forced A/B runs at pass and iteration level (tools/select_sweep.sh, tools/slots_ab.sh;
            // profiles/r01_slots_ab.txt, r01_select_sweep_*.txt):
            //   float32 from 32 padded factors, float64 from 64: whenever a wave would otherwise re-fetch a wide B - up to
            //     40 slots (+6..+59 % it/s at 10k x 5k .. 20k x 20k), or more slots if the partial tiles stay below ~12 %
            //     of the X bytes (1000 x 120000 x 64: 129 slots, 174 us vs 320 us) - on contractions that are not short;
            //   float64 up to 32 factors: the 4x4x4 kernel is the faster stream; its fixed rounds lose to the stream-K
            //     balancing at 8 slots only on long contractions (20k x 20k: +7.6 % it/s; 2500 x 20000: -5 %);
            //   16-factor float32 and short contractions (448 rows: 20 us vs 25 us; 3008: 26 us vs 34 us at 33 slots) keep
            //     the small-shard kernel.
            auto use_ct = [&](int sl, int64_t K) -> bool {
                const char* e = getenv("LCX_CT_MAX_SLOTS");
                if (e && *e) return sl <= atoi(e);
                if (sl <= 6) return true;
                const bool small_partials = sl <= 160 && (double)sl * Mp <= 0.12 * (double)K;
                if (sizeof(T) == 4) return CT >= 2 && K >= 4096 && (sl <= 40 || small_partials);
                if (CT >= 4) return K >= 4096 && (sl <= 40 || small_partials);
                return K >= 8192 && sl <= 8;
            };
            // The occupancy that sizes a stream-K grid belongs to the instantiation that will be launched (gemm_cr / gemm_ct, panel-major
            // or row-major operand), which depends on the layout, which depends on whether BOTH passes take the stream-K kernels: decide
            // with the panel instantiations first (unless LCX_X_LAYOUT=rows forbids the layout), and if the shard does not end up
            // panel-major redo the geometry of its stream-K passes for the row-major ones.
            const char* lay = getenv("LCX_X_LAYOUT");
            const bool rows_only = lay && !strcmp(lay, "rows"), force_panel = lay && !strcmp(lay, "panel");
            auto stream_k = [&](bool as_panel, bool decide) {
                int nb, ns, sl;
                // X.B^T: gemm_cr on the panel copy / the row-major X (single-copy mode), gemm_ct on the transposed copy
                ct_geometry<T, CT>(cus, h->ldx, h->Npad, env_int("LCX_CT_NB", 0), &nb, &ns, &sl, as_panel, as_panel || h->single_copy);
                if (decide) h->nt_ct = force_panel || h->single_copy || (force ? !strcmp(force, "ct") : use_ct(sl, h->ldx));
                if (h->nt_ct) { h->nt_nb = nb; h->nt_nsuper = ns; h->nt_S = sl; h->nt_KW = ct_kw<T, CT>(as_panel); }
                ct_geometry<T, CT>(cus, h->Npad, h->ldx, env_int("LCX_CT_NB", 0), &nb, &ns, &sl, as_panel, false);
                if (decide) h->tn_ct = force_panel || (force ? !strcmp(force, "ct") : use_ct(sl, h->Npad));
                if (h->tn_ct) { h->tn_nb = nb; h->tn_nsuper = ns; h->tn_S = sl; h->tn_KW = ct_kw<T, CT>(as_panel); }
            };
            stream_k(!rows_only, true);
            // Both passes on the stream-K kernels: ONE panel-major copy of the shard serves both at full speed (gemm_kernels.hpp,
            // PanelW) - no transposed copy, half the resident bytes.  LCX_X_LAYOUT=rows keeps the row-major layout(s), =panel forces
            // the stream-K kernels and the panel layout on any shape.
            h->panel = h->nt_ct && h->tn_ct && !rows_only;
            if (h->panel) h->single_copy = false;
            else if (!rows_only && (h->nt_ct || h->tn_ct)) stream_k(false, false);
        }
        // merged pass: float32, 32 / 64 padded factors, large shards (the 2 Mp-wide gemm_ct does the flops of both passes at
        // a higher rate and reads X once); LCX_MERGED_PASS=0 turns it off
        h->merged_ok = false;
        if constexpr (sizeof(T) == 4 && CT >= 2 && CT <= 4) {
            if (h->nt_ct && env_int("LCX_MERGED_PASS", 1) != 0) {
                int nb, ns, sl;
                ct_geometry<T, 2 * CT>(cus, h->ldx, h->Npad, env_int("LCX_CT_NB", 0), &nb, &ns, &sl, h->panel, h->panel || h->single_copy);
                if (sl <= 8) { h->merged_ok = true; h->nt2_nb = nb; h->nt2_nsuper = ns; h->nt2_S = sl; }
            }
        }
        {
            // contraction splits of the m x m Gram launches (Y^T.Y, W.W^T, H): up to 128, aiming at 12 waves per CU (round 4: with 64 /
            // a quarter of that the launches of the large shards left CUs idle - config 3: 65 + 59 -> 38 + 34 us, config-4 shard
            // 124 + 121 -> 70 + 86 us, the slot reductions 5 us dearer); short contractions are capped by work as before
            const int gdiv = env_int("LCX_GRAM_WAVES_DIV", 1), gcap = env_int("LCX_GRAM_CAP", 128);
            h->gn_S = pick_split(Mp / (16 * Geo<T, CT>::G_RT), pick_kw(kgn), kgn, h->target_waves / (gdiv > 0 ? gdiv : 1), gcap);
            h->gv_S = pick_split(Mp / (16 * Geo<T, CT>::G_RT), pick_kw(kgv), kgv, h->target_waves / (gdiv > 0 ? gdiv : 1), gcap);
        }
        int64_t groups = cdiv(h->V, VPB);
        h->pv_grid = (int)(groups < 1024 ? (groups < 1 ? 1 : groups) : 1024);
        // many slots (few column tiles): one wide reduction after the pass instead of a long serial sum per element in
        // every consumer
        h->tn_slots = (h->tn_S >= WIDE_SPLITS && cdiv(h->ldx * Mp, 32) < (1 << 20)) ? 1 : h->tn_S;
        return LCX_OK;
    }

    // ---- Gram matrix of a [K][Mp] array (K multiple of 16): partials -> gpart[S][Mp][Mp] ------
    static int gram(lcx_ctx* h, const T* A, int64_t K, const T* scale, int S, const int* skip, T* dst) {
        if constexpr (WIDE) {
            if (scale) return wide_gemm<true, true>(h, A, Mp, A, Mp, scale, dst, Mp, Mp, Mp, K, S, skip);
            return wide_gemm<true, false>(h, A, Mp, A, Mp, nullptr, dst, Mp, Mp, Mp, K, S, skip);
        } else {
            const int kw = pick_kw(K / 16);
            constexpr int RT = Geo<T, CT>::G_RT;
            if (scale)
                return launch_tn<T, CT, RT, true>(h->stream, A, Mp, K, Mp, A, scale, dst, S, kw, skip);
            return launch_tn<T, CT, RT, false>(h->stream, A, Mp, K, Mp, A, nullptr, dst, S, kw, skip);
        }
    }

    // Y(_partial) = X . B^T (linearcorex.py:247 / :210) as a contraction over the rows of XT
    // the pass alone: partial slots -> dst[slot][Npad][Mp].  rows >= 0: only the output rows [r0, r0 + rows) (a multiple of the
    // row tile) - the wave-split kernels only, whose per-tile contraction split does not depend on the grid (same bits as the
    // whole launch)
    static int nt_pass(lcx_ctx* h, const T* B, const int* skip, T* dst, int64_t r0 = 0, int64_t rows = -1) {
        if constexpr (WIDE) {
            (void)r0; (void)rows;
            LCXCHECK((wide_gemm<false, false>(h, P<T>(h->X), h->ldx, B, Mp, nullptr, dst, Mp, h->Npad, Mp, h->ldx, 1, skip)));
        } else {
            if (rows >= 0) {
                if (h->panel || h->single_copy || h->nt_ct) return fail(LCX_ERR_STATE, "row chunks of the pass exist for the wave-split kernels only");
                if (h->f64_4x4) {
                    if constexpr (sizeof(T) == 8 && CT <= 2)
                        LCXCHECK((launch_tn4<CT>(h->stream, P<double>(h->XT) + r0, h->Npad, h->ldx, rows, (const double*)B, (double*)dst + r0 * Mp, h->nt_S,
                                                 h->nt_KW, skip, h->Npad)));
                } else
                    LCXCHECK((launch_tn<T, CT, Geo<T, CT>::TN_RT, false, false>(h->stream, P<T>(h->XT) + r0, h->Npad, h->ldx, rows, B, nullptr,
                                                                                 dst + r0 * Mp, h->nt_S, h->nt_KW, skip, h->Npad)));
                return LCX_OK;
            }
            if (h->panel)
                LCXCHECK((launch_cr<T, CT, true>(h->stream, P<T>(h->X), h->Npad * PanelW<T>::v, h->ldx, h->Npad, B, dst, h->nt_nb, h->nt_nsuper,
                                                 h->nt_S, skip, h->split ? h->bsp : nullptr, h->n_cus)));
            else if (h->single_copy)
                LCXCHECK((launch_cr<T, CT>(h->stream, P<T>(h->X), h->ldx, h->ldx, h->Npad, B, dst, h->nt_nb, h->nt_nsuper, h->nt_S, skip)));
            else if (h->nt_ct)
                LCXCHECK((launch_ct<T, CT>(h->stream, P<T>(h->XT), h->Npad, h->ldx, h->Npad, B, dst, h->nt_nb, h->nt_nsuper, h->nt_S, skip)));
            else if (h->f64_4x4) {
                if constexpr (sizeof(T) == 8 && CT <= 2)
                    LCXCHECK((launch_tn4<CT>(h->stream, P<double>(h->XT), h->Npad, h->ldx, h->Npad, (const double*)B, (double*)dst, h->nt_S, h->nt_KW, skip)));
            } else
                LCXCHECK((launch_tn<T, CT, Geo<T, CT>::TN_RT, false, false>(h->stream, P<T>(h->XT), h->Npad, h->ldx, h->Npad, B, nullptr,
                                                                             dst, h->nt_S, h->nt_KW, skip)));
        }
        return LCX_OK;
    }
    // sum the partial slots of the elements [e0, e0 + n) of Y into ybuf (and `also`): the kernel - and so the order of every
    // element's sum - depends on the slot count alone, so a row chunk gets the bits of the whole reduction
    static int nt_reduce(lcx_ctx* h, const int* skip, T* also, int64_t e0, int64_t n) {
        const int64_t ntot = h->Npad * Mp;
        const bool wide = h->nt_S >= WIDE_SPLITS && cdiv(ntot, 32) < (1 << 20);
        if (wide) {
            hipLaunchKernelGGL((reduce_partials_wide_kernel<T, T>), dim3((unsigned)cdiv(n, 32)), dim3(256), 0,
                               h->stream, P<T>(h->ypart) + e0, h->nt_S, n, ntot, P<T>(h->ybuf) + e0, skip, also ? also + e0 : also);
            KCHECK();
        } else if (h->nt_S > 1) {
            hipLaunchKernelGGL((reduce_partials_kernel<T, T>), dim3((unsigned)(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024)), dim3(256), 0,
                               h->stream, P<T>(h->ypart) + e0, h->nt_S, n, ntot, P<T>(h->ybuf) + e0, skip, also ? also + e0 : also);
            KCHECK();
        }
        return LCX_OK;
    }
    static int nt_big(lcx_ctx* h, const void* Bv, const int* skip, bool with_bj = false, T* also = nullptr) {
        const T* B = reinterpret_cast<const T*>(Bv);
        TimingPair tp;
        LCXCHECK(timing_begin(h, 0, &tp));
        T* dst = h->nt_S > 1 ? P<T>(h->ypart) : P<T>(h->ybuf);
        LCXCHECK(nt_pass(h, B, skip, dst));
        LCXCHECK(timing_end(h, 0, &tp));
        const int64_t n = h->Npad * Mp;
        const bool wide = h->nt_S >= WIDE_SPLITS && cdiv(n, 32) < (1 << 20);
        if (with_bj) {
            // partial tiles of Y (if split) and the Bj partials of grad_kernel, one launch
            if (wide) {
                const int yblocks = (int)cdiv(n, 32);
                hipLaunchKernelGGL((reduce_y_bj_kernel<T, true>), dim3(yblocks + Mp), dim3(PV_THREADS), 0, h->stream, P<T>(h->ypart),
                                   h->nt_S, n, P<T>(h->ybuf), yblocks, h->bjpart, h->pv_grid, Mp, P<T>(h->ybuf) + n);
            } else {
                const int yblocks = h->nt_S > 1 ? (int)(cdiv(n, PV_THREADS) < 1024 ? cdiv(n, PV_THREADS) : 1024) : 0;
                hipLaunchKernelGGL((reduce_y_bj_kernel<T, false>), dim3(yblocks + Mp), dim3(PV_THREADS), 0, h->stream, P<T>(h->ypart),
                                   h->nt_S, n, P<T>(h->ybuf), yblocks, h->bjpart, h->pv_grid, Mp, P<T>(h->ybuf) + n);
            }
            KCHECK();
        } else {
            LCXCHECK(nt_reduce(h, skip, also, 0, n));
        }
        return LCX_OK;
    }

    // ---- LCX_Y_PIPELINE=chunks: the N x m all-reduces of lcx_moments_a ([Y_partial | W.W^T partial]) and lcx_update_b ([Y_g partial |
    // Bj partial], separate-pass form) in row chunks on a second stream ----
    // Chunk c of the summed Y is all-reduced as soon as its slot reduction has run, while the main stream goes on with chunk c+1:
    // with the wave-split kernels (small shards, where the exchange is exposed: DESIGN.md section 6) the PASS itself is launched
    // per row chunk, so the all-reduce of chunk c overlaps the pass of chunk c+1; with the stream-K kernels the pass is one launch
    // and only the reductions overlap.  Every element is summed over slots and over ranks exactly as without chunks (two ranks:
    // bit-identical; more ranks: the transport's order within a call may depend on the element's position in the call).  The
    // W.W^T tail sits right behind Y in the buffer and rides in the last chunk.  Every rank issues the same chunks in the same order.
    static int ypipe_init(lcx_ctx* h) {
        if (h->comm_stream) return LCX_OK;
        HIPCHECK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        for (auto& e : h->ypipe_ev) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return LCX_OK;
    }
    // B = the weights of the evaluated set (lcx_moments_a: tail = W.W^T partial) or, with_bj, the gradient (lcx_update_b: tail = the
    // Bj partial sums of grad_kernel, :302)
    static int y_pass_pipelined(lcx_ctx* h, const T* w, bool with_bj) {
        LCXCHECK(ypipe_init(h));
        int64_t tile = 64;
        bool chunk_pass = false;
        if constexpr (!WIDE) {
            chunk_pass = !(h->panel || h->single_copy || h->nt_ct);
            if (chunk_pass) tile = 16 * Geo<T, CT>::TN_RT;
        }
        const int64_t tiles = h->Npad / tile;
        const int C = (int)(h->ypipe < tiles ? h->ypipe : tiles);
        // A row chunk of the pass is a launch of tiles / C x nt_S blocks: worth it only while that still fills one round of resident
        // blocks - the whole launch is sized to exactly that (single_round_split), so a quarter of it leaves three quarters of the chip
        // idle and the pass, HBM-bound, takes about as long per chunk as in one piece (config 2: 157 tiles x 3 splits = 471 blocks
        // on 512 slots).  Otherwise the pass stays one launch and only the slot reductions and all-reduces go out in chunks
        // ("chunks:n:pass" forces the per-chunk pass: tests).
        if (chunk_pass && !h->ypipe_force_pass && (tiles / C) * (int64_t)h->nt_S < (int64_t)h->n_cus * h->nt_bpc) chunk_pass = false;
        // the tail first: it rides in the last chunk's all-reduce
        if (with_bj) {
            hipLaunchKernelGGL((reduce_y_bj_kernel<T, false>), dim3(Mp), dim3(PV_THREADS), 0, h->stream, P<T>(h->ypart), 1, h->Npad * Mp,
                               P<T>(h->ybuf), 0, h->bjpart, h->pv_grid, Mp, P<T>(h->ybuf) + h->Npad * Mp);
            KCHECK();
        } else {
            LCXCHECK(gram_w(h, w));
        }
        TimingPair tp;
        LCXCHECK(timing_begin(h, 0, &tp));
        T* dst = h->nt_S > 1 ? P<T>(h->ypart) : P<T>(h->ybuf);
        if (!chunk_pass) {
            LCXCHECK(nt_pass(h, w, nullptr, dst));
            LCXCHECK(timing_end(h, 0, &tp));
        }
        for (int c = 0; c < C; ++c) {
            const int64_t r0 = tiles * c / C * tile, r1 = tiles * (c + 1) / C * tile;
            if (chunk_pass) {
                LCXCHECK(nt_pass(h, w, nullptr, dst, r0, r1 - r0));
                if (c == C - 1) LCXCHECK(timing_end(h, 0, &tp));
            }
            LCXCHECK(nt_reduce(h, nullptr, (T*)nullptr, r0 * Mp, (r1 - r0) * Mp));
            HIPCHECK(hipEventRecord(h->ypipe_ev[c], h->stream));
            HIPCHECK(hipStreamWaitEvent(h->comm_stream, h->ypipe_ev[c], 0));
            const int64_t count = (r1 - r0) * Mp + (c == C - 1 ? (int64_t)Mp * Mp : 0);
            LCXCHECK(exchange_on(h, h->comm_stream, P<T>(h->ybuf) + r0 * Mp, count, DT));
        }
        HIPCHECK(hipEventRecord(h->ypipe_ev[16], h->comm_stream));
        HIPCHECK(hipStreamWaitEvent(h->stream, h->ypipe_ev[16], 0));
        return LCX_OK;
    }
    static int make_xt(lcx_ctx* h) {
        if (h->single_copy || h->panel) { HIPCHECK(hipStreamSynchronize(h->stream)); return LCX_OK; }
        dim3 grid((unsigned)(h->ldx / 64), (unsigned)(h->Npad / 64));
        hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, h->stream, P<T>(h->X), h->ldx, P<T>(h->XT), h->Npad);
        KCHECK();
        HIPCHECK(hipStreamSynchronize(h->stream));
        return LCX_OK;
    }
    static int tn_big(lcx_ctx* h, const int* skip) {
        TimingPair tp;
        LCXCHECK(timing_begin(h, 1, &tp));
        if constexpr (WIDE) {
            LCXCHECK((wide_gemm<true, false>(h, P<T>(h->X), h->ldx, P<T>(h->ybuf), Mp, nullptr, P<T>(h->dpart), Mp, h->ldx, Mp, h->Npad, 1, skip)));
        } else {
            if (h->panel)
                LCXCHECK((launch_ct<T, CT, true>(h->stream, P<T>(h->X), h->Npad * PanelW<T>::v, h->Npad, h->ldx, P<T>(h->ybuf), P<T>(h->dpart),
                                                 h->tn_nb, h->tn_nsuper, h->tn_S, skip, h->split ? h->bsp : nullptr, h->n_cus)));
            else if (h->tn_ct)
                LCXCHECK((launch_ct<T, CT>(h->stream, P<T>(h->X), h->ldx, h->Npad, h->ldx, P<T>(h->ybuf), P<T>(h->dpart), h->tn_nb, h->tn_nsuper,
                                           h->tn_S, skip)));
            else if (h->f64_4x4) {
                if constexpr (sizeof(T) == 8 && CT <= 2)
                    LCXCHECK((launch_tn4<CT>(h->stream, P<double>(h->X), h->ldx, h->Npad, h->ldx, P<double>(h->ybuf), P<double>(h->dpart), h->tn_S,
                                             h->tn_KW, skip)));
            } else
                LCXCHECK((launch_tn<T, CT, Geo<T, CT>::TN_RT, false, false>(h->stream, P<T>(h->X), h->ldx, h->Npad, h->ldx, P<T>(h->ybuf), nullptr,
                                                                             P<T>(h->dpart), h->tn_S, h->tn_KW, skip)));
        }
        LCXCHECK(timing_end(h, 1, &tp));
        if (h->tn_slots != h->tn_S) {
            const int64_t n = h->ldx * Mp;
            hipLaunchKernelGGL((reduce_partials_wide_kernel<T, T>), dim3((unsigned)cdiv(n, 32)), dim3(256), 0, h->stream,
                               P<T>(h->dpart), h->tn_S, n, n, P<T>(h->dpart), skip, (T*)nullptr);      // in place: slot 0
            KCHECK();
        }
        return LCX_OK;
    }

    // W.W^T partials of the shard -> gpartw; with several ranks also summed into the ybuf tail, which
    // is what gets all-reduced
    static int gram_w(lcx_ctx* h, const T* w) {
        LCXCHECK(gram(h, w, h->ldx, nullptr, h->gv_S, nullptr, P<T>(h->gpartw)));
        if (h->exchange) {
            hipLaunchKernelGGL((reduce_wide_kernel<T, T>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                               P<T>(h->gpartw), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp,
                               P<T>(h->ybuf) + h->Npad * Mp, (const int*)nullptr);
            KCHECK();
        }
        return LCX_OK;
    }

    static int moments_a(lcx_ctx* h, int which) {
        if (which == 1 && h->y1_ready && use_merged(h)) {       // the merged pass of lcx_update_b left it in ybuf / set 1
            h->y1_ready = false;
            return LCX_OK;
        }
        if (which == 1 && h->yk_ready) {                        // trial_by_linearity left it in ybuf (and the W'.W'^T tail, if exchanged)
            h->yk_ready = false;
            return LCX_OK;
        }
        h->y1_ready = h->yk_ready = false;
        T* w = P<T>(h->Wt[which]);
        if (h->exchange && h->ypipe > 1 && h->tr.kind != 0) return y_pass_pipelined(h, w, false);
        // without an exchange the summed Y is final: the set's own copy is written by the same reduction
        LCXCHECK(nt_big(h, w, nullptr, false, (!h->exchange && h->nt_S > 1) ? P<T>(h->set[which].Y) : (T*)nullptr));
        if (!h->exchange) return LCX_OK;         // nothing to exchange: W.W^T is formed with Y^T.Y in lcx_moments_b (one launch)
        LCXCHECK(gram_w(h, w));
        return exchange(h, h->ybuf, h->ybuf_main, DT);           // L1 of SURVEY 8e: [Y_partial | W.W^T partial]
    }

    // per-factor moments; ysrc != null: first form the Y^T.Y partials of that Y
    static int small(lcx_ctx* h, int which, double eps, int quick, const T* ysrc) {
        MomentSet& s = h->set[which];
        if (ysrc) {
            if (!h->exchange) LCXCHECK(gram_pair(h, P<T>(h->Wt[which]), ysrc));
            else LCXCHECK(gram(h, ysrc, h->Npad, nullptr, h->gn_S, nullptr, P<T>(h->gpart)));
        }
        SmallDesc sd{s.uj, s.ry, s.wmag};
        const T* gw = h->exchange ? P<T>(h->ybuf) + h->Npad * Mp : P<T>(h->gpartw);
        hipLaunchKernelGGL((small_moments_kernel<T>), dim3(Mp * Mp / 32), dim3(256), 0, h->stream, P<T>(h->gpart),
                           h->gn_S, gw, h->exchange ? 1 : h->gv_S, Mp, h->M, h->Ndiv, eps, quick, sd, s.st,
                           h->ticket);
        KCHECK();
        return LCX_OK;
    }

    // W'^T-Gram and Y'^T-Gram of a trial in one launch
    static int gram_pair(lcx_ctx* h, const T* w, const T* y) {
        if constexpr (WIDE) {
            LCXCHECK(gram(h, w, h->ldx, nullptr, h->gv_S, nullptr, P<T>(h->gpartw)));
            LCXCHECK(gram(h, y, h->Npad, nullptr, h->gn_S, nullptr, P<T>(h->gpart)));
        } else {
            LCXCHECK(gram_pair_tuned(h, w, y));
        }
        if (h->exchange) {
            hipLaunchKernelGGL((reduce_wide_kernel<T, T>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                               P<T>(h->gpartw), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp,
                               P<T>(h->ybuf) + h->Npad * Mp, (const int*)nullptr);
            KCHECK();
        }
        return LCX_OK;
    }
    static int gram_pair_tuned(lcx_ctx* h, const T* w, const T* y) {
        constexpr int RT = Geo<T, CT>::G_RT;
        const int kgv = (int)(h->ldx / 16), kgn = (int)(h->Npad / 16);
        // With several ranks the Y^T.Y Gram feeds uj / TC, which every rank must form bit-identically (the line-search
        // decisions are taken from them): its wave split may then depend on the replicated n_samples only, never on the
        // local shard width (448 vs 512 local columns would pick 2 vs 4 waves and sum in a different order).
        const int kw = h->exchange ? pick_kw(kgn) : pick_kw(kgv < kgn ? kgv : kgn);
        GramProblem<T> p0{w, P<T>(h->gpartw), kgv, h->gv_S}, p1{y, P<T>(h->gpart), kgn, h->gn_S};
        dim3 grid((unsigned)(Mp / (16 * RT)), (unsigned)(h->gv_S > h->gn_S ? h->gv_S : h->gn_S), 2);
        const size_t lds = (size_t)kw * 16 * RT * Mp * sizeof(T);
        if (lds > 48 * 1024) {
            switch (kw) {
                case 1: LCXCHECK(allow_lds(gram_pair_kernel<T, CT, RT, 1>, lds)); break;
                case 2: LCXCHECK(allow_lds(gram_pair_kernel<T, CT, RT, 2>, lds)); break;
                default: LCXCHECK(allow_lds(gram_pair_kernel<T, CT, RT, 4>, lds)); break;
            }
        }
        switch (kw) {
            case 1: hipLaunchKernelGGL((gram_pair_kernel<T, CT, RT, 1>), grid, dim3(64), lds, h->stream, p0, p1); break;
            case 2: hipLaunchKernelGGL((gram_pair_kernel<T, CT, RT, 2>), grid, dim3(128), lds, h->stream, p0, p1); break;
            default: hipLaunchKernelGGL((gram_pair_kernel<T, CT, RT, 4>), grid, dim3(256), lds, h->stream, p0, p1); break;
        }
        KCHECK();
        return LCX_OK;
    }

    static int epilogue(lcx_ctx* h, int which, double eps, bool linear, double eta) {
        MomentSet& s = h->set[which];
        const int* skip = &s.st->invalid;
        bool on_mfma = false;
        if constexpr (sizeof(T) == 4 && (Mp == 64 || Mp == 128)) {
            // the m x m operator product on the matrix pipe, a wave per 16 variables (moment_kernels.hpp, PvMfma); LCX_PV_MFMA=0:
            // the thread-per-(variable, factor) form
            if (pv_mfma()) {
                const size_t lds = PvMfma<Mp>::lds_bytes;
                LCXCHECK(allow_lds(moments_epilogue_mfma_kernel<Mp>, lds));
                hipLaunchKernelGGL((moments_epilogue_mfma_kernel<Mp>), dim3(h->pv_grid), dim3(64 * PvMfma<Mp>::NW), lds, h->stream,
                                   P<float>(h->dpart), h->tn_slots, h->ldx * Mp,
                                   linear ? P<float>(h->set[0].D) : (const float*)nullptr, P<float>(h->ddir), (float)eta, P<float>(s.D),
                                   P<float>(h->Wt[which]), s.ry, h->V, h->Ndiv, eps,
                                   P<float>(s.rho), P<float>(s.rir), P<float>(s.qij), P<float>(s.si), P<float>(s.q2), P<float>(s.hscale),
                                   h->tcpart, skip);
                on_mfma = true;
            }
        }
        if (!on_mfma) {
            const size_t lds = ((size_t)(OpInLds<Mp>::v ? Mp * Mp : 0) + (size_t)VPB * Mp) * sizeof(T);
            LCXCHECK(allow_lds(moments_epilogue_kernel<T, Mp>, lds));
            hipLaunchKernelGGL((moments_epilogue_kernel<T, Mp>), dim3(h->pv_grid), dim3(NTV), lds, h->stream,
                               P<T>(h->dpart), h->tn_slots, h->ldx * Mp,
                               linear ? P<T>(h->set[0].D) : (const T*)nullptr, P<T>(h->ddir), (T)eta, P<T>(s.D),
                               P<T>(h->Wt[which]), s.ry, h->V, h->Ndiv, eps,
                               P<T>(s.rho), P<T>(s.rir), P<T>(s.qij), P<T>(s.si), P<T>(s.q2), P<T>(s.hscale),
                               h->tcpart, skip);
        }
        KCHECK();
        // H partial of THIS set (:294), so that the update that follows an accepted trial needs no exchange of
        // its own: it rides in the scalar all-reduce of the evaluation.  The same launch carries the tail block of the
        // evaluation (log sums -> sbuf[0..1], a pending update_tangent -> sbuf[2], TC + publication with one GPU).
        const int single = !h->exchange;
        const unsigned int seq = single ? ++h->seq_next : 0u;
        TcTail tail{h->tcpart, h->pv_grid, h->tanpart, h->tan_blocks, h->sbuf, s.st, h->set[0].st, s.hst_dev, seq, single, skip};
        h->tan_blocks = 0;
        if constexpr (WIDE) {
            // the tail of the evaluation first (the host sees TC as early as possible), then the H Gram on gemm_wide
            hipLaunchKernelGGL((tc_tail_kernel<T>), dim3(1), dim3(256), 0, h->stream, tail);
            KCHECK();
            LCXCHECK(gram(h, P<T>(s.rir), h->ldx, P<T>(s.hscale), h->gv_S, skip, P<T>(h->gpart)));
        } else {
            constexpr int RT = Geo<T, CT>::G_RT;
            const int kgroups = (int)(h->ldx / 16), kw = pick_kw(kgroups);
            dim3 grid((unsigned)(Mp / (16 * RT)), (unsigned)h->gv_S, 2);
            const size_t glds = (size_t)kw * 16 * RT * Mp * sizeof(T);
            if (glds > 48 * 1024) {
                switch (kw) {
                    case 1: LCXCHECK(allow_lds(gram_tc_kernel<T, CT, RT, 1>, glds)); break;
                    case 2: LCXCHECK(allow_lds(gram_tc_kernel<T, CT, RT, 2>, glds)); break;
                    default: LCXCHECK(allow_lds(gram_tc_kernel<T, CT, RT, 4>, glds)); break;
                }
            }
            switch (kw) {
                case 1: hipLaunchKernelGGL((gram_tc_kernel<T, CT, RT, 1>), grid, dim3(64), glds, h->stream, P<T>(s.rir), P<T>(s.hscale), P<T>(h->gpart), kgroups, h->gv_S, skip, tail); break;
                case 2: hipLaunchKernelGGL((gram_tc_kernel<T, CT, RT, 2>), grid, dim3(128), glds, h->stream, P<T>(s.rir), P<T>(s.hscale), P<T>(h->gpart), kgroups, h->gv_S, skip, tail); break;
                default: hipLaunchKernelGGL((gram_tc_kernel<T, CT, RT, 4>), grid, dim3(256), glds, h->stream, P<T>(s.rir), P<T>(s.hscale), P<T>(h->gpart), kgroups, h->gv_S, skip, tail); break;
            }
            KCHECK();
        }
        if (single) s.seq_expect = seq;
        hipLaunchKernelGGL((reduce_wide_kernel<T, double>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                           P<T>(h->gpart), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp, h->sbuf + SB_H, skip);
        KCHECK();
        return exchange(h, h->sbuf, SB_H + Mp * Mp, LCX_F64);    // L2 + L3 + L5: TC sums, pending tangent, H of this set
    }

    static int moments_b(lcx_ctx* h, int which, double eps, int quick) {
        MomentSet& s = h->set[which];
        if (which == 0) h->spec_dirty = false;          // this evaluation leaves the H of set 0 in sbuf
        // keep the (all-reduced) Y of this set: the linear trial mode starts from it
        if (h->exchange || h->nt_S == 1)         // otherwise lcx_moments_a already wrote it
            HIPCHECK(hipMemcpyAsync(s.Y, h->ybuf, (size_t)h->Npad * Mp * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
        LCXCHECK(small(h, which, eps, quick, P<T>(h->ybuf)));
        LCXCHECK(tn_big(h, &s.st->invalid));
        return epilogue(h, which, eps, false, 0.0);
    }

    // ---- linear trial mode: moments of ws + eta*update without touching X ----------------------
    // a: w_update (:320), its W.W^T partial -> ybuf tail, and Y' = Y + eta*Y(update) -> set 1
    static int trial_linear_a(lcx_ctx* h, double eta) {
        const int64_t n1 = h->V * Mp, n2 = h->Npad * Mp;
        hipLaunchKernelGGL((axpy2_kernel<T>), dim3((unsigned)(cdiv(n1 + n2, 256) < 2048 ? cdiv(n1 + n2, 256) : 2048)), dim3(256), 0,
                           h->stream, P<T>(h->Wt[0]), P<T>(h->update), P<T>(h->Wt[1]), n1,
                           P<T>(h->set[0].Y), P<T>(h->ydir), P<T>(h->set[1].Y), n2, (T)eta);
        KCHECK();
        LCXCHECK(gram_pair(h, P<T>(h->Wt[1]), P<T>(h->set[1].Y)));
        return exchange(h, P<T>(h->ybuf) + h->Npad * Mp, (int64_t)Mp * Mp, DT);      // W'.W'^T partial
    }
    // b: (W.W^T tail global) uj, flag, D' = D + eta*D(update), rho ... TC partial sums
    static int trial_linear_b(lcx_ctx* h, double eps, double eta) {
        LCXCHECK(small(h, 1, eps, 1, nullptr));
        return epilogue(h, 1, eps, true, eta);
    }

    static int moments_c(lcx_ctx* h, int which) {
        if (!h->exchange) return LCX_OK;             // the epilogue already published
        MomentSet& s = h->set[which];
        const unsigned int seq = ++h->seq_next;
        hipLaunchKernelGGL((tc_final_kernel<T>), dim3(1), dim3(1), 0, h->stream, h->sbuf, s.st, s.hst_dev, seq);
        KCHECK();
        s.seq_expect = seq;
        return LCX_OK;
    }

    static int update_a(lcx_ctx* h) {
        MomentSet& s = h->set[0];
        LCXCHECK(gram(h, P<T>(s.rir), h->ldx, P<T>(s.hscale), h->gv_S, nullptr, P<T>(h->gpart)));
        hipLaunchKernelGGL((reduce_wide_kernel<T, double>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                           P<T>(h->gpart), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp, h->sbuf + SB_H, (const int*)nullptr);
        KCHECK();
        return exchange(h, h->sbuf + SB_H, (int64_t)Mp * Mp, LCX_F64);
    }

    // grad (:296-300) and the per-block Bj partials (:302) of set `which`, from its moments and the H its evaluation left in sbuf
    static int launch_grad(lcx_ctx* h, int which) {
        MomentSet& s = h->set[which];
        if constexpr (sizeof(T) == 4 && (Mp == 64 || Mp == 128)) {
            if (pv_mfma()) {
                const size_t lds = PvMfma<Mp>::lds_bytes;
                LCXCHECK(allow_lds(grad_mfma_kernel<Mp>, lds));
                hipLaunchKernelGGL((grad_mfma_kernel<Mp>), dim3(h->pv_grid), dim3(64 * PvMfma<Mp>::NW), lds, h->stream, P<float>(h->Wt[which]),
                                   P<float>(s.rho), P<float>(s.rir), P<float>(s.qij), P<float>(s.si), P<float>(s.q2), s.uj, h->sbuf + SB_H, h->V,
                                   P<float>(h->grad), h->bjpart, use_merged(h) ? P<float>(h->gw) : (float*)nullptr);
                KCHECK();
                return LCX_OK;
            }
        }
        const size_t lds = ((size_t)(OpInLds<Mp>::v ? Mp * (Mp + 1) : 0) + (size_t)VPB * Mp) * sizeof(T) + (size_t)VPB * Mp * sizeof(double) + 8;
        LCXCHECK(allow_lds(grad_kernel<T, Mp>, lds));
        hipLaunchKernelGGL((grad_kernel<T, Mp>), dim3(h->pv_grid), dim3(NTV), lds, h->stream, P<T>(h->Wt[which]),
                           P<T>(s.rho), P<T>(s.rir), P<T>(s.qij), P<T>(s.si), P<T>(s.q2), s.uj, h->sbuf + SB_H, h->V,
                           P<T>(h->grad), h->bjpart, use_merged(h) ? P<T>(h->gw) : (T*)nullptr);
        KCHECK();
        return LCX_OK;
    }
    static int update_b(lcx_ctx* h, double eps) {
        (void)eps;
        if (h->spec_dirty) {                            // an abandoned speculation overwrote the H of set 0 in sbuf
            if (!self_contained(h)) return fail(LCX_ERR_STATE, "abandoned lcx_iterate speculation while the caller owns the exchange");
            LCXCHECK(update_a(h));
            h->spec_dirty = false;
        }
        MomentSet& s = h->set[0];
        const bool merged = use_merged(h);
        if (h->grad_ready) h->grad_ready = false;       // lcx_iterate already computed it behind the accepted trial's evaluation
        else LCXCHECK(launch_grad(h, 0));
        if (!merged) {
            if (h->exchange && h->ypipe > 1 && h->tr.kind != 0) return y_pass_pipelined(h, P<T>(h->grad), true);
            LCXCHECK(nt_big(h, P<T>(h->grad), nullptr, true));
            return exchange(h, h->ybuf, h->ybuf_main, DT);       // L4: [Y_g partial | Bj partial]
        }
        if constexpr (sizeof(T) == 4 && CT >= 2 && CT <= 4) {
            // Bj (:302) does not wait for the pass: sum its per-block partials now, form update and ws + update (:303, :320) and
            // put both B operands side by side, then ONE pass over X for [Y_g | Y of the first trial]
            const int64_t n = h->Npad * Mp;
            hipLaunchKernelGGL((reduce_y_bj_kernel<T, false>), dim3(Mp), dim3(PV_THREADS), 0, h->stream, P<T>(h->ypart), 1, n,
                               P<T>(h->ybuf), 0, h->bjpart, h->pv_grid, Mp, P<T>(h->ybuf) + n);
            KCHECK();
            const T* bj = P<T>(h->ybuf) + n;
            if (h->exchange) {
                // several ranks: `update` needs Bj over ALL variables before the pass - one tiny all-reduce in front of it - and
                // the tail of ybuf is needed again for W'.W'^T of the first trial, so the global Bj moves to a buffer of its own
                LCXCHECK(exchange(h, P<T>(h->ybuf) + n, Mp, DT));
                HIPCHECK(hipMemcpyAsync(h->bjg, P<T>(h->ybuf) + n, sizeof(T) * Mp, hipMemcpyDeviceToDevice, h->stream));
                bj = P<T>(h->bjg);
            }
            const int grid = update_grid(h);
            hipLaunchKernelGGL((update_kernel<T, Mp>), dim3(grid), dim3(PV_THREADS), 0, h->stream, P<T>(h->dpart), 0, h->ldx * Mp,
                               P<T>(h->grad), P<T>(h->Wt[0]), s.uj, bj, h->V, h->Ndiv, eps, P<T>(h->update),
                               P<T>(h->sgrad), h->tanpart, (const T*)nullptr, (T*)nullptr, grid, (const T*)nullptr, (const T*)nullptr,
                               (int64_t)0, (T*)nullptr, P<T>(h->Wt[1]), h->world, P<T>(h->gw), 0);
            KCHECK();
            TimingPair tp;
            LCXCHECK(timing_begin(h, 2, &tp));
            if (h->panel)
                LCXCHECK((launch_cr<T, 2 * CT, true>(h->stream, P<T>(h->X), h->Npad * PanelW<T>::v, h->ldx, h->Npad, P<T>(h->gw), P<T>(h->y2part),
                                                     h->nt2_nb, h->nt2_nsuper, h->nt2_S, nullptr, h->split ? h->bsp : nullptr, h->n_cus)));
            else if (h->single_copy)
                LCXCHECK((launch_cr<T, 2 * CT>(h->stream, P<T>(h->X), h->ldx, h->ldx, h->Npad, P<T>(h->gw), P<T>(h->y2part), h->nt2_nb,
                                               h->nt2_nsuper, h->nt2_S, nullptr)));
            else
                LCXCHECK((launch_ct<T, 2 * CT>(h->stream, P<T>(h->XT), h->Npad, h->ldx, h->Npad, P<T>(h->gw), P<T>(h->y2part), h->nt2_nb,
                                               h->nt2_nsuper, h->nt2_S, nullptr)));
            LCXCHECK(timing_end(h, 2, &tp));
            const int64_t n2 = 2 * n;
            hipLaunchKernelGGL((reduce_split_kernel<T>), dim3((unsigned)(cdiv(n2, 256) < 2048 ? cdiv(n2, 256) : 2048)), dim3(256), 0,
                               h->stream, P<T>(h->y2part), h->nt2_S, n2, Mp, P<T>(h->ygbuf), P<T>(h->ybuf), P<T>(h->set[1].Y));
            KCHECK();
            if (h->exchange) {
                // what lcx_moments_a(1) would have left for the first trial - W'.W'^T partial in the tail - and ONE all-reduce of
                // [Y' | W'.W'^T | Y_g] (ygbuf sits right behind the tail); lcx_moments_b copies the summed Y' into set 1
                LCXCHECK(gram_w(h, P<T>(h->Wt[1])));
                LCXCHECK(exchange(h, h->ybuf, h->ybuf_elems, DT));
            }
            h->w1_ready = h->y1_ready = true;
        }
        return LCX_OK;
    }
    // the merged pass needs: its buffers, the Y-space tangent, and - with several ranks - the exchange inside the library (a caller
    // that all-reduces the buffers itself between the levels does not know about the Bj exchange in front of the pass)
    static bool use_merged(const lcx_ctx* h) {
        return h->merged_ok && (!h->exchange || h->tr.kind != 0) && !h->full_sig && h->gw != nullptr;
    }
    static int update_grid(const lcx_ctx* h) { return (int)(cdiv(h->V * Mp, PV_THREADS) < 1536 ? cdiv(h->V * Mp, PV_THREADS) : 1536); }

    static int update_c(lcx_ctx* h, double eps) {
        MomentSet& s = h->set[0];
        // The second pass of _sig (X^T.Y_g, :211) only feeds update_tangent, which is available in Y space after
        // the first pass (see update_kernel); it is run when the linear trial mode needs D(update) as well.
        if (h->full_sig) LCXCHECK(tn_big(h, nullptr));
        const int grid = update_grid(h);
        const int64_t ny = h->Npad * Mp;
        const int gridy = (int)(cdiv(ny, PV_THREADS) < 512 ? cdiv(ny, PV_THREADS) : 512);
        if (h->y1_ready && use_merged(h)) {
            // merged flow: update / ws + update were formed before the pass (lcx_update_b); what is left is the Y-space part -
            // Y(update) and the Y term of update_tangent - from Y_g, which sits in its own buffer
            hipLaunchKernelGGL((update_kernel<T, Mp>), dim3(gridy), dim3(PV_THREADS), 0, h->stream, P<T>(h->dpart), 0, h->ldx * Mp,
                               P<T>(h->grad), P<T>(h->Wt[0]), s.uj, h->exchange ? P<T>(h->bjg) : P<T>(h->ybuf) + h->Npad * Mp, h->V,
                               h->Ndiv, eps, P<T>(h->update),
                               P<T>(h->sgrad), h->tanpart, (const T*)nullptr, (T*)nullptr, 0, P<T>(h->ygbuf), P<T>(s.Y), ny, P<T>(h->ydir),
                               (T*)nullptr, h->world, (T*)nullptr, grid);
            KCHECK();
            h->tan_blocks = grid + gridy;
            return LCX_OK;
        }
        hipLaunchKernelGGL((update_kernel<T, Mp>), dim3(grid + gridy), dim3(PV_THREADS), 0, h->stream, P<T>(h->dpart),
                           h->full_sig ? h->tn_slots : 0, h->ldx * Mp, P<T>(h->grad), P<T>(h->Wt[0]), s.uj, P<T>(h->ybuf) + h->Npad * Mp, h->V,
                           h->Ndiv, eps, P<T>(h->update), P<T>(h->sgrad), h->tanpart,
                           h->full_sig ? P<T>(s.D) : (const T*)nullptr, h->full_sig ? P<T>(h->ddir) : (T*)nullptr, grid, P<T>(h->ybuf),
                           P<T>(s.Y), ny, P<T>(h->ydir), P<T>(h->Wt[1]), h->world);
        KCHECK();
        h->tan_blocks = grid + gridy;        // summed into sbuf[2] / the state by the tail of the first trial's evaluation
        h->w1_ready = true;
        return LCX_OK;
    }

    static int make_trial(lcx_ctx* h, double eta) {
        if (eta == 1.0 && h->w1_ready) return LCX_OK;        // update_kernel already wrote ws + update
        h->y1_ready = false;
        const int64_t n = h->V * Mp;
        hipLaunchKernelGGL((axpy_kernel<T>), dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0,
                           h->stream, P<T>(h->Wt[0]), P<T>(h->update), (T)eta, n, P<T>(h->Wt[1]));
        KCHECK();
        return LCX_OK;
    }

    // A back-tracking trial after the first one (:320-321 at eta = 1/2, 1/4, ...): w_update = ws + eta update, and - X.u^T being
    // linear in u - X.w_update^T = Y + eta X.update^T, where Y belongs to the current solution and X.update^T = -rj (Y_g - c Y)
    // is what lcx_update_c formed from this iteration's exact pass X.grad^T.  The trial then needs ONE pass over X (X^T.Y') instead of
    // two.  Nothing is carried across iterations except the Y of an accepted solution, which every such step mixes with fresh
    // products in a convex combination: no drift, no re-anchoring (unlike the linear trial mode, which also reuses X^T.Y).
    static int trial_by_linearity(lcx_ctx* h, double eta) {
        const int64_t n1 = h->V * Mp, n2 = h->Npad * Mp;
        hipLaunchKernelGGL((axpy2_kernel<T>), dim3((unsigned)(cdiv(n1 + n2, 256) < 2048 ? cdiv(n1 + n2, 256) : 2048)), dim3(256), 0,
                           h->stream, P<T>(h->Wt[0]), P<T>(h->update), P<T>(h->Wt[1]), n1,
                           P<T>(h->set[0].Y), P<T>(h->ydir), P<T>(h->ybuf), n2, (T)eta);
        KCHECK();
        if (h->exchange) {
            // Y and X.update^T are already sums over all ranks; only W'.W'^T of the trial is still per shard
            LCXCHECK(gram_w(h, P<T>(h->Wt[1])));
            LCXCHECK(exchange(h, P<T>(h->ybuf) + h->Npad * Mp, (int64_t)Mp * Mp, DT));
        } else if (h->nt_S > 1) {
            HIPCHECK(hipMemcpyAsync(h->set[1].Y, h->ybuf, (size_t)n2 * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
        }
        h->w1_ready = h->y1_ready = false;
        h->yk_ready = true;
        return LCX_OK;
    }

    // ---- one whole fixed-point iteration with its back-tracking line search (:290-334), one GPU -------------------
    // :321 for the weights in set 1, and right behind it the gradient those weights would need next (:296-300): if the trial
    // is accepted that gradient is already there when the host has decided, if not it is overwritten by the next trial's
    static int evaluate_trial(lcx_ctx* h, double eps) {
        LCXCHECK(moments_a(h, 1));
        LCXCHECK(moments_b(h, 1, eps, 1));
        LCXCHECK(moments_c(h, 1));                   // several ranks: TC / tangent from the summed scalars, publication
        h->early_grad = false;
        // Worth it while the gradient kernel is shorter than the host's decision latency (~20 us): up to ~1M (variable, factor)
        // pairs (config 2: 5 us).  On large shards a rejected trial would waste more than the gap it hides (config 4 shard: 289 us).
        // (The linear trial mode keeps grad / sig_grad of the direction in flight.)
        if (!h->full_sig && h->V * (int64_t)Mp <= ((int64_t)1 << 20)) {
            LCXCHECK(launch_grad(h, 1));
            h->early_grad = true;
        }
        return LCX_OK;
    }
    static int direction_and_trial(lcx_ctx* h, double eps) {
        LCXCHECK(update_b(h, eps));                  // grad (:296-300), Y_g = X.grad^T (:210), Bj (:302)
        LCXCHECK(update_c(h, eps));                  // update (:303), update_tangent partials (:305), ws + update
        h->have_direction = true;
        LCXCHECK(make_trial(h, 1.0));                // :320 at eta = 1 (update_kernel wrote it already)
        return evaluate_trial(h, eps);
    }
    static int iterate(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out) {
        const int rc = iterate_body(h, eps, tol, tc_cur, more, out);
        if (rc != LCX_OK) {
            // a failure part-way (a launch error, a publication that never arrived) leaves sbuf with the H of some trial and the
            // direction / trial flags half set: make the level API start over (lcx_update_b then restores the H of set 0)
            h->spec_pending = false;
            h->spec_dirty = true;
            h->early_grad = h->grad_ready = h->have_direction = h->w1_ready = h->y1_ready = h->yk_ready = false;
        }
        return rc;
    }
    static int iterate_body(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out) {
        if (!self_contained(h))
            return fail(LCX_ERR_STATE, "lcx_iterate with several ranks needs the exchange inside the library (lcx_comm_init or "
                                       "lcx_set_exchange_hook); otherwise the caller exchanges between the levels "
                                       "(lcx_update_b ... lcx_moments_c)");
        const bool consumed = h->spec_pending && h->spec_eps == eps;
        if (h->spec_pending && !consumed) cancel_speculation(h);
        h->spec_pending = false;
        if (!consumed) LCXCHECK(direction_and_trial(h, eps));
        double eta = 1.0, tangent = 0.0, last_tc = __builtin_nan("");
        int trials = 0, invalid_trials = 0, too_small = 0;
        bool first = true, have_last = false, last_invalid = false;
        const double eta_min = tol < 1e-10 ? tol : 1e-10;                       // :316
        while (true) {
            if (!first) {
                if (eta < eta_min) { too_small = 1; break; }                     // :316-319
                if (h->reuse_y && !h->full_sig) LCXCHECK(trial_by_linearity(h, eta));
                else LCXCHECK(make_trial(h, eta));                               // :320
                LCXCHECK(evaluate_trial(h, eps));                                // :321
            }
            ++trials;
            LCXCHECK(wait_published(h, h->set[1]));
            const SetState st = *h->set[1].hst;
            if (first) {
                first = false;
                tangent = st.tangent;                                            // :305, summed by the first trial's tail
                if (tangent >= 0) {                                              // :306-311: keep ws, discard the trial
                    LCXCHECK(update_a(h));                                       // its H went to sbuf: restore set 0's
                    h->early_grad = h->grad_ready = false;
                    h->have_direction = false;
                    h->w1_ready = h->y1_ready = false;
                    out[0] = 1; out[1] = tc_cur; out[2] = tangent; out[3] = trials - 1; out[4] = 0; out[5] = 0; out[6] = trials; out[7] = 0;
                    return LCX_OK;
                }
            }
            have_last = true;
            last_invalid = st.invalid != 0;
            last_tc = st.tc;
            if (last_invalid) { ++invalid_trials; eta *= 0.5; continue; }        // :322-326
            if (!(-last_tc <= -tc_cur + 0.1 * eta * tangent)) { eta *= 0.5; continue; }   // :327-332
            break;
        }
        // self.ws, self.moments = w_update, m_update (:139, :334)
        h->w1_ready = h->y1_ready = false;
        std::swap(h->Wt[0], h->Wt[1]);
        std::swap(h->set[0], h->set[1]);
        h->have_direction = false;
        const bool ok = have_last && !last_invalid;
        h->grad_ready = ok && h->early_grad && !too_small;   // the gradient of the accepted trial = the next iteration's gradient
        h->early_grad = false;
        int speculated = 0;
        if (ok && more) {
            const double delta = last_tc > tc_cur ? last_tc - tc_cur : tc_cur - last_tc;
            if (!(delta < tol)) {             // the caller will iterate again (:152): get the GPU going before it asks
                LCXCHECK(direction_and_trial(h, eps));
                h->spec_pending = true;
                h->spec_eps = eps;
                speculated = 1;
            }
        }
        out[0] = ok ? 0 : 2; out[1] = last_tc; out[2] = tangent; out[3] = trials; out[4] = invalid_trials; out[5] = too_small;
        out[6] = trials; out[7] = speculated;
        return LCX_OK;
    }

    static int rescale(lcx_ctx* h, double e0, double e1) {
        const int64_t n = h->V * Mp;
        hipLaunchKernelGGL((rescale_kernel<T>), dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0,
                           h->stream, P<T>(h->Wt[0]), n, Mp, h->set[0].uj, h->set[0].wmag, e0, e1);
        KCHECK();
        return LCX_OK;
    }

    static int init_scale(lcx_ctx* h) {
        LCXCHECK(small(h, 0, 0.0, 0, P<T>(h->ybuf)));
        const int64_t n = h->V * Mp;
        hipLaunchKernelGGL((init_scale_kernel<T>), dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0,
                           h->stream, P<T>(h->Wt[0]), n, Mp, h->M, h->set[0].uj);
        KCHECK();
        return LCX_OK;
    }

    static int permute(lcx_ctx* h, const int32_t* order) {
        HIPCHECK(hipMemcpyAsync(h->order_dev, order, sizeof(int) * h->M, hipMemcpyHostToDevice, h->stream));
        const int64_t n = h->V * Mp;
        hipLaunchKernelGGL((permute_kernel<T>), dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0,
                           h->stream, P<T>(h->Wt[0]), P<T>(h->Wt[1]), n, Mp, h->M, h->order_dev);
        KCHECK();
        HIPCHECK(hipStreamSynchronize(h->stream));
        std::swap(h->Wt[0], h->Wt[1]);
        return LCX_OK;
    }

    // detail sums; optionally materialise MI / XiZj / Xi2|Y into scratch arrays
    static int detail(lcx_ctx* h, int which, T* mi_o, T* xz_o, T* x2y_o) {
        MomentSet& s = h->set[which];
        hipLaunchKernelGGL(invert_kernel, dim3(1), dim3(WIDE ? 1024 : 256), 0, h->stream, s.ry, Mp, h->invwork, h->ryinv);
        KCHECK();
        const size_t lds = ((size_t)(OpInLds<Mp>::v ? Mp * Mp : 0) + (size_t)VPB * Mp) * sizeof(T) + (size_t)VPB * Mp * sizeof(double) + 8;
        LCXCHECK(allow_lds(detail_kernel<T, Mp>, lds));
        hipLaunchKernelGGL((detail_kernel<T, Mp>), dim3(h->pv_grid), dim3(NTV), lds, h->stream, P<T>(s.rho),
                           h->ryinv, h->V, h->M, mi_o, xz_o, x2y_o, h->detpart);
        KCHECK();
        hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(h->M + 3), dim3(PV_THREADS), 0, h->stream, h->detpart, h->pv_grid,
                           h->M + 3, h->sbuf + sb_det(Mp), (const int*)nullptr);
        KCHECK();
        return LCX_OK;
    }

    // ---- synergistic branch (discourage_overlap=False; :336-384) -------------------------------------
    static int syn_alloc(lcx_ctx* h) {
        for (int k = 0; k < 2; ++k) {
            MomentSet& s = h->set[k];
            if (s.xz) continue;
            HIPCHECK(hipMalloc(&s.xz, (size_t)h->ldx * Mp * sizeof(T)));
            HIPCHECK(hipMalloc(&s.x2y, (size_t)h->ldx * sizeof(T)));
            HIPCHECK(hipMalloc((void**)&s.cy, sizeof(double) * Mp * Mp));
            HIPCHECK(hipMalloc((void**)&s.yj2, sizeof(double) * Mp));
            HIPCHECK(hipMalloc((void**)&s.inv_sd, sizeof(double) * Mp));
            HIPCHECK(hipMemsetAsync(s.xz, 0, (size_t)h->ldx * Mp * sizeof(T), h->stream));
            HIPCHECK(hipMemsetAsync(s.x2y, 0, (size_t)h->ldx * sizeof(T), h->stream));
        }
        return LCX_OK;
    }
    // b: (ybuf = global Y) cy, Y_j^2, ry (:356-358), X^T.Y (:355), rho (:359), X_i Z_j (:367), X_i^2|Y (:368) and
    //    the per-shard sums behind TCs / additivity / TC (:371-373) -> sbuf[0 .. m+3)
    static int syn_moments_b(lcx_ctx* h, int which, double yscale) {
        LCXCHECK(syn_alloc(h));
        MomentSet& s = h->set[which];
        LCXCHECK(gram(h, P<T>(h->ybuf), h->Npad, nullptr, h->gn_S, nullptr, P<T>(h->gpart)));
        hipLaunchKernelGGL((syn_small_kernel<T>), dim3(1), dim3(256), 0, h->stream, P<T>(h->gpart), h->gn_S, Mp, h->M, h->Ndiv,
                           yscale, s.cy, s.yj2, s.ry, s.inv_sd, s.st);
        KCHECK();
        LCXCHECK(tn_big(h, nullptr));
        const int64_t total = h->V * Mp;
        hipLaunchKernelGGL((syn_rho_kernel<T>), dim3((unsigned)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048)), dim3(256), 0, h->stream,
                           P<T>(h->dpart), h->tn_slots, h->ldx * Mp, total, Mp, h->Ndiv, s.inv_sd, P<T>(s.D), P<T>(s.rho));
        KCHECK();
        hipLaunchKernelGGL(invert_kernel, dim3(1), dim3(WIDE ? 1024 : 256), 0, h->stream, s.ry, Mp, h->invwork, h->ryinv);
        KCHECK();
        const size_t lds = ((size_t)(OpInLds<Mp>::v ? Mp * Mp : 0) + (size_t)VPB * Mp) * sizeof(T) + (size_t)VPB * Mp * sizeof(double) + 8;
        LCXCHECK(allow_lds(detail_kernel<T, Mp>, lds));
        // X_i Z_j = solve(cy, X_i Y_j^T)^T = (ry^-1 rho)_j / sd_j ; X_i^2|Y = 1 - rho^T ry^-1 rho ; hscale <- 1 / X_i^2|Y
        hipLaunchKernelGGL((detail_kernel<T, Mp>), dim3(h->pv_grid), dim3(NTV), lds, h->stream, P<T>(s.rho), h->ryinv, h->V, h->M,
                           (T*)nullptr, P<T>(s.xz), P<T>(s.x2y), h->detpart, (const double*)s.inv_sd, P<T>(s.hscale));
        KCHECK();
        hipLaunchKernelGGL((sum_partials_kernel<double>), dim3(h->M + 3), dim3(PV_THREADS), 0, h->stream, h->detpart, h->pv_grid,
                           h->M + 3, h->sbuf + sb_det(Mp), (const int*)nullptr);
        KCHECK();
        return exchange(h, h->sbuf + sb_det(Mp), h->M + 3, LCX_F64);
    }
    static int syn_moments_c(lcx_ctx* h, int which) {
        MomentSet& s = h->set[which];
        const unsigned int seq = ++h->seq_next;
        hipLaunchKernelGGL((syn_tc_kernel<T>), dim3(1), dim3(1), 0, h->stream, h->sbuf + sb_det(Mp), h->M, s.st, s.hst_dev, seq);
        KCHECK();
        s.seq_expect = seq;
        return LCX_OK;
    }
    // H partial (:378) -> sbuf[0 .. Mp^2)
    static int syn_update_a(lcx_ctx* h) {
        MomentSet& s = h->set[0];
        if (!s.xz) return fail(LCX_ERR_STATE, "lcx_syn_update_a before lcx_syn_moments_b");
        LCXCHECK(gram(h, P<T>(s.xz), h->ldx, P<T>(s.hscale), h->gv_S, nullptr, P<T>(h->gpart)));
        hipLaunchKernelGGL((reduce_wide_kernel<T, double>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                           P<T>(h->gpart), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp, h->sbuf + SB_H, (const int*)nullptr);
        KCHECK();
        return exchange(h, h->sbuf + SB_H, (int64_t)Mp * Mp, LCX_F64);
    }
    // ws' = (1-eta) ws + eta (R - H ws) (:380-382) -> set 1
    static int syn_update_b(lcx_ctx* h, double eta) {
        MomentSet& s = h->set[0];
        const size_t lds = ((size_t)(OpInLds<Mp>::v ? Mp * (Mp + 1) : 0) + (size_t)VPB * Mp) * sizeof(T);
        LCXCHECK(allow_lds(syn_update_kernel<T, Mp>, lds));
        hipLaunchKernelGGL((syn_update_kernel<T, Mp>), dim3(h->pv_grid), dim3(NTV), lds, h->stream, P<T>(h->Wt[0]), P<T>(s.xz),
                           P<T>(s.hscale), h->sbuf + SB_H, h->V, (T)eta, P<T>(h->Wt[1]));
        KCHECK();
        return LCX_OK;
    }
    // get_covariance rows [row0, row0 + nrows) -> out_host (row-major, leading dimension ld_out elements); see CovStage
    // Staging shared by get_covariance and predict.  Every resource is created on its own guard: a call that failed half
    // way (say a locked-memory limit on the pinned blocks) leaves the stage retryable instead of half built.
    static int cov_stage(lcx_ctx* h, bool need_op_a, bool need_op_b) {
        if (!h->cov) h->cov = new CovStage();
        CovStage& c = *h->cov;
        const size_t mv = (size_t)h->ldx * Mp * sizeof(T);
        auto dmalloc = [&](void** p, size_t bytes) -> int {
            if (*p) return LCX_OK;
            HIPCHECK(hipMalloc(p, bytes));
            h->bytes_resident += bytes;
            return LCX_OK;
        };
        if (!c.copy_stream) HIPCHECK(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            if (!c.ev_k[k]) HIPCHECK(hipEventCreateWithFlags(&c.ev_k[k], hipEventDisableTiming));
            if (!c.ev_c[k]) HIPCHECK(hipEventCreateWithFlags(&c.ev_c[k], hipEventDisableTiming));
            if (!c.t_a[k]) HIPCHECK(hipEventCreate(&c.t_a[k]));
            if (!c.t_b[k]) HIPCHECK(hipEventCreate(&c.t_b[k]));
        }
        LCXCHECK(dmalloc(&c.std_dev, (size_t)h->ldx * sizeof(T)));
        LCXCHECK(dmalloc(&c.mean_dev, (size_t)h->ldx * sizeof(T)));
        if (need_op_a) LCXCHECK(dmalloc(&c.op_a, mv));
        if (need_op_b) LCXCHECK(dmalloc(&c.op_b, mv));
        if (!c.block_bytes) {
            const int64_t ldo = h->ldx;
            int64_t rows = (int64_t)(64u << 20) / (ldo * (int64_t)sizeof(T)) / 64 * 64;
            if (rows < 64) rows = 64;
            const int64_t cap = round_up(h->V, 64) > 8192 ? round_up(h->V, 64) : 8192;     // covariance blocks never exceed V rows
            if (rows > cap) rows = cap;
            c.block_rows = rows;
            c.block_bytes = (size_t)rows * ldo * sizeof(T);
        }
        for (int k = 0; k < 2; ++k) {
            LCXCHECK(dmalloc(&c.dev[k], c.block_bytes));
            if (!c.pin[k]) HIPCHECK(hipHostMalloc(&c.pin[k], c.block_bytes, hipHostMallocDefault));
        }
        return LCX_OK;
    }
    static void place_rows(const T* src, int64_t src_ld, T* dst, int64_t dst_ld, int64_t rows, int64_t cols) {
        // pinned block -> the caller's matrix; the destination is usually freshly allocated pageable memory, i.e. this is
        // where its pages are first touched: a few threads keep it off the critical path of the PCIe copies
        const int64_t bytes = rows * cols * (int64_t)sizeof(T);
        static const int max_threads = []() {
            const char* e = getenv("LCX_HOST_THREADS");
            int n = (e && *e) ? atoi(e) : 12;
            return n < 1 ? 1 : (n > 64 ? 64 : n);
        }();
        const int nt = bytes >= (8 << 20) ? max_threads : 1;
        auto work = [=](int t) {
            const int64_t r0 = rows * t / nt, r1 = rows * (t + 1) / nt;
            if (src_ld == cols && dst_ld == cols) memcpy(dst + r0 * cols, src + r0 * cols, (size_t)(r1 - r0) * cols * sizeof(T));
            else for (int64_t r = r0; r < r1; ++r) memcpy(dst + r * dst_ld, src + r * src_ld, (size_t)cols * sizeof(T));
        };
        if (nt == 1) { work(0); return; }
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    static int covariance_blocks(lcx_ctx* h, bool syn, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out_host,
                                 int64_t ld_out, double* kernel_seconds) {
        MomentSet& s = h->set[0];
        LCXCHECK(cov_stage(h, !syn, syn));
        CovStage& c = *h->cov;
        const int64_t V = h->V, ldo = h->ldx;
        const int64_t brows = c.block_rows;
        HIPCHECK(hipMemcpyAsync(c.std_dev, std_host, sizeof(T) * V, hipMemcpyHostToDevice, h->stream));
        const int64_t n = h->ldx * Mp;
        const unsigned pg = (unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048);
        if (syn)
            hipLaunchKernelGGL((cov_prep_kernel<T>), dim3(pg), dim3(256), 0, h->stream, (const T*)nullptr, (const T*)nullptr, P<T>(s.D), n, Mp,
                               (T)(1.0 / h->Ndiv), (T*)nullptr, P<T>(c.op_b));
        else
            hipLaunchKernelGGL((cov_prep_kernel<T>), dim3(pg), dim3(256), 0, h->stream, P<T>(s.rir), P<T>(s.si), (const T*)nullptr, n, Mp, (T)0,
                               P<T>(c.op_a), (T*)nullptr);
        KCHECK();
        // The destination is usually a freshly allocated, never touched NumPy array: its first-touch page faults are
        // what the end-to-end time of a large matrix is made of.  Ask for huge pages on the page-aligned interior of the
        // borrowed buffer (a hint; ignored where transparent huge pages are off).
        {
            const size_t bytes = (size_t)nrows * (size_t)ld_out * sizeof(T);
            if (bytes >= ((size_t)8 << 20)) {
                const uintptr_t pg = (uintptr_t)2 << 20;
                const uintptr_t a0 = ((uintptr_t)out_host + pg - 1) & ~(pg - 1), a1 = ((uintptr_t)out_host + bytes) & ~(pg - 1);
                if (a1 > a0) (void)madvise((void*)a0, (size_t)(a1 - a0), MADV_HUGEPAGE);
            }
        }
        const T* opa = syn ? P<T>(s.xz) : P<T>(c.op_a);
        const T* opb = syn ? P<T>(c.op_b) : P<T>(c.op_a);
        const T denom = syn ? (T)1 : (T)(1.0 - eps * eps);
        T* out = P<T>(out_host);
        const int64_t nblk = cdiv(nrows, brows);
        double ksec = 0.0;
        for (int64_t k = 0; k <= nblk; ++k) {
            if (k < nblk) {
                const int b = (int)(k & 1);
                const int64_t r0 = row0 + k * brows, nr = (nrows - k * brows) < brows ? (nrows - k * brows) : brows;
                dim3 grid((unsigned)cdiv(V, 64), (unsigned)cdiv(nr, 64));
                HIPCHECK(hipEventRecord(c.t_a[b], h->stream));
                hipLaunchKernelGGL((cov_syrk_kernel<T, Mp, false>), grid, dim3(256), 0, h->stream, opa, opb, P<T>(c.std_dev), V, r0, nr, denom,
                                   P<T>(c.dev[b]), ldo, (const T*)nullptr, 0);
                KCHECK();
                HIPCHECK(hipEventRecord(c.t_b[b], h->stream));
                HIPCHECK(hipEventRecord(c.ev_k[b], h->stream));
                HIPCHECK(hipStreamWaitEvent(c.copy_stream, c.ev_k[b], 0));
                HIPCHECK(hipMemcpy2DAsync(c.pin[b], (size_t)V * sizeof(T), c.dev[b], (size_t)ldo * sizeof(T), (size_t)V * sizeof(T), (size_t)nr,
                                          hipMemcpyDeviceToHost, c.copy_stream));
                HIPCHECK(hipEventRecord(c.ev_c[b], c.copy_stream));
            }
            if (k >= 1) {
                const int b = (int)((k - 1) & 1);
                const int64_t nr = (nrows - (k - 1) * brows) < brows ? (nrows - (k - 1) * brows) : brows;
                HIPCHECK(hipEventSynchronize(c.ev_c[b]));
                {
                    float ms = 0.f;
                    HIPCHECK(hipEventElapsedTime(&ms, c.t_a[b], c.t_b[b]));
                    ksec += (double)ms * 1e-3;
                }
                place_rows(P<T>(c.pin[b]), V, out + (k - 1) * brows * ld_out, ld_out, nr, V);
            }
        }
        c.last_kernel_seconds = ksec;
        if (kernel_seconds) *kernel_seconds = ksec;
        return LCX_OK;
    }
    static int covariance_syn(lcx_ctx* h, const void* std_host, int64_t row0, int64_t nrows, void* out_host) {
        MomentSet& s = h->set[0];
        if (!s.xz) return fail(LCX_ERR_STATE, "synergistic covariance needs lcx_syn_moments_b first");
        return covariance_blocks(h, true, 0.0, std_host, row0, nrows, out_host, h->V, nullptr);
    }

    // [Vp][Mp] device -> (m, V) or (V, m) host
    static int fetch_mv(lcx_ctx* h, const T* dev, T* host, bool as_m_by_v) {
        std::vector<T> tmp((size_t)h->V * Mp);
        HIPCHECK(hipMemcpyAsync(tmp.data(), dev, tmp.size() * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        for (int64_t v = 0; v < h->V; ++v)
            for (int j = 0; j < h->M; ++j) {
                if (as_m_by_v) host[(int64_t)j * h->V + v] = tmp[v * Mp + j];
                else host[v * h->M + j] = tmp[v * Mp + j];
            }
        return LCX_OK;
    }
    static int fetch_v(lcx_ctx* h, const T* dev, T* host) {
        HIPCHECK(hipMemcpyAsync(host, dev, (size_t)h->V * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        return LCX_OK;
    }
    static int fetch_small(lcx_ctx* h, const double* dev, int rows, int cols, T* host) {
        std::vector<double> tmp((size_t)Mp * Mp);
        HIPCHECK(hipMemcpyAsync(tmp.data(), dev, sizeof(double) * (rows == 1 ? Mp : Mp * Mp), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        for (int a = 0; a < rows; ++a)
            for (int b = 0; b < cols; ++b) host[a * cols + b] = (T)tmp[(rows == 1 ? 0 : a * Mp) + b];
        return LCX_OK;
    }

    static int get_moment(lcx_ctx* h, int which, int key, double eps, void* out) {
        (void)eps;
        MomentSet& s = h->set[which];
        T* o = P<T>(out);
        T* scr = P<T>(h->scratch);
        const int64_t n = h->ldx * Mp;
        switch (key) {
            case LCX_M_UJ: return fetch_small(h, s.uj, 1, h->M, o);
            case LCX_M_RY: return fetch_small(h, s.ry, h->M, h->M, o);
            case LCX_M_H: {
                LCXCHECK(fetch_small(h, h->sbuf + SB_H, h->M, h->M, o));
                for (int a = 0; a < h->M; ++a) o[a * h->M + a] = (T)0;
                return LCX_OK;
            }
            case LCX_M_RHO: return fetch_mv(h, P<T>(s.rho), o, true);
            case LCX_M_RHOINVRHO: return fetch_mv(h, P<T>(s.rir), o, true);
            case LCX_M_QIJ: return fetch_mv(h, P<T>(s.qij), o, true);
            case LCX_M_INVRHO: {
                hipLaunchKernelGGL((invrho_kernel<T>), dim3(1024), dim3(256), 0, h->stream, P<T>(s.rho), n, scr);
                KCHECK();
                return fetch_mv(h, scr, o, true);
            }
            case LCX_M_SI: return fetch_v(h, P<T>(s.si), o);
            case LCX_M_QISI2: return fetch_v(h, P<T>(s.q2), o);
            case LCX_M_MI: LCXCHECK(detail(h, which, scr, nullptr, nullptr)); return fetch_mv(h, scr, o, true);
            case LCX_M_XIZJ: LCXCHECK(detail(h, which, nullptr, scr, nullptr)); return fetch_mv(h, scr, o, false);
            case LCX_M_XI2_GIVEN_Y: LCXCHECK(detail(h, which, nullptr, nullptr, scr)); return fetch_v(h, scr, o);
            case LCX_M_GRAD: return fetch_mv(h, P<T>(h->grad), o, true);
            case LCX_M_UPDATE: return fetch_mv(h, P<T>(h->update), o, true);
            case LCX_M_SIG_GRAD: return fetch_mv(h, P<T>(h->sgrad), o, true);
            case LCX_M_Y: {
                std::vector<T> tmp((size_t)h->N * Mp);
                HIPCHECK(hipMemcpyAsync(tmp.data(), h->ybuf, tmp.size() * sizeof(T), hipMemcpyDeviceToHost, h->stream));
                HIPCHECK(hipStreamSynchronize(h->stream));
                for (int64_t r = 0; r < h->N; ++r)
                    for (int j = 0; j < h->M; ++j) o[r * h->M + j] = tmp[r * Mp + j];
                return LCX_OK;
            }
            case LCX_M_SYN_XIZJ: if (!s.xz) return fail(LCX_ERR_STATE, "no synergistic moments"); return fetch_mv(h, P<T>(s.xz), o, false);
            case LCX_M_SYN_X2Y: if (!s.xz) return fail(LCX_ERR_STATE, "no synergistic moments"); return fetch_v(h, P<T>(s.x2y), o);
            case LCX_M_SYN_XIYJ: {
                LCXCHECK(fetch_mv(h, P<T>(s.D), o, false));
                for (int64_t i = 0; i < h->V * h->M; ++i) o[i] = o[i] / (T)h->Ndiv;
                return LCX_OK;
            }
            case LCX_M_CY: if (!s.xz) return fail(LCX_ERR_STATE, "no synergistic moments"); return fetch_small(h, s.cy, h->M, h->M, o);
            case LCX_M_YJ2: if (!s.xz) return fail(LCX_ERR_STATE, "no synergistic moments"); return fetch_small(h, s.yj2, 1, h->M, o);
            default: return fail(LCX_ERR_ARG, "unknown moment key");
        }
    }

    static int set_moment(lcx_ctx* h, int which, int key, const void* in) {
        MomentSet& s = h->set[which];
        const T* src = reinterpret_cast<const T*>(in);
        if (key == LCX_M_SI) {
            HIPCHECK(hipMemcpyAsync(s.si, src, sizeof(T) * h->V, hipMemcpyHostToDevice, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            return LCX_OK;
        }
        if (key == LCX_M_SYN_XIZJ || key == LCX_M_SYN_XIYJ) {       // (nv, m) host arrays of the synergistic branch (:453)
            LCXCHECK(syn_alloc(h));
            std::vector<T> tmp((size_t)h->ldx * Mp, (T)0);
            const T scale = key == LCX_M_SYN_XIYJ ? (T)h->Ndiv : (T)1;  // kept as X^T.Y = N * X_i Y_j (:355)
            for (int64_t v = 0; v < h->V; ++v)
                for (int j = 0; j < h->M; ++j) tmp[v * Mp + j] = src[v * h->M + j] * scale;
            HIPCHECK(hipMemcpyAsync(key == LCX_M_SYN_XIZJ ? s.xz : s.D, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            return LCX_OK;
        }
        if (key != LCX_M_RHOINVRHO) return fail(LCX_ERR_ARG, "lcx_set_moment: only RHOINVRHO, SI, SYN_XIZJ and SYN_XIYJ can be restored");
        std::vector<T> tmp((size_t)h->ldx * Mp, (T)0);
        for (int j = 0; j < h->M; ++j)
            for (int64_t v = 0; v < h->V; ++v) tmp[v * Mp + j] = src[(int64_t)j * h->V + v];
        HIPCHECK(hipMemcpyAsync(s.rir, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        return LCX_OK;
    }

    static int set_ws(lcx_ctx* h, const void* w_host) {
        const T* w = reinterpret_cast<const T*>(w_host);
        std::vector<T> tmp((size_t)h->ldx * Mp, (T)0);
        for (int j = 0; j < h->M; ++j)
            for (int64_t v = 0; v < h->V; ++v) tmp[v * Mp + j] = w[(int64_t)j * h->V + v];
        HIPCHECK(hipMemcpyAsync(h->Wt[0], tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        return LCX_OK;
    }

    static int covariance_full(lcx_ctx* h, int syn, double eps, const void* std_host, void* out_host, int64_t ld_out, double* ksec) {
        if (syn && !h->set[0].xz) return fail(LCX_ERR_STATE, "synergistic covariance needs lcx_syn_moments_b first");
        return covariance_blocks(h, syn != 0, eps, std_host, 0, h->V, out_host, ld_out, ksec);
    }
    static int covariance(lcx_ctx* h, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out_host) {
        return covariance_blocks(h, false, eps, std_host, row0, nrows, out_host, h->V, nullptr);
    }

    // theta = (mean, std) of the working dtype -> the stage's device vectors (kind 0: unused)
    static int stage_theta(lcx_ctx* h, int kind, const void* mean_h, const void* std_h) {
        CovStage& c = *h->cov;
        if (kind == PP_KIND_NONE) return LCX_OK;
        HIPCHECK(hipMemcpyAsync(c.mean_dev, mean_h, sizeof(T) * h->V, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(hipMemcpyAsync(c.std_dev, std_h, sizeof(T) * h->V, hipMemcpyHostToDevice, h->stream));
        return LCX_OK;
    }

    // predict (:440-441): out (n_rows x V) = invert(y . X_i Z_j^T), produced in row blocks like get_covariance: the
    // rank-Mp product + the inverse marginal map on the device, two pinned staging blocks, the host-side placement of
    // block k-1 under the kernel of block k+1 and the copy of block k.
    static int predict(lcx_ctx* h, const void* y_host, int64_t n_rows, int syn, const void* xz_host, int kind, const void* mean_h,
                       const void* std_h, void* out_host, int64_t ld_out, double* kernel_seconds) {
        MomentSet& s = h->set[0];
        LCXCHECK(cov_stage(h, true, false));
        CovStage& c = *h->cov;
        const int64_t V = h->V, ldo = h->ldx, brows = c.block_rows;
        LCXCHECK(stage_theta(h, kind, mean_h, std_h));
        // operand B = X_i Z_j [Vp][Mp]
        const T* xz = nullptr;
        if (xz_host) {                                            // restored model: the caller's (V x m) matrix
            std::vector<T> tmp((size_t)h->ldx * Mp, (T)0);
            const T* src = reinterpret_cast<const T*>(xz_host);
            for (int64_t v = 0; v < V; ++v)
                for (int j = 0; j < h->M; ++j) tmp[v * Mp + j] = src[v * h->M + j];
            HIPCHECK(hipMemcpyAsync(c.op_a, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            xz = P<T>(c.op_a);
        } else if (syn) {
            if (!s.xz) return fail(LCX_ERR_STATE, "lcx_predict: no synergistic moments resident (lcx_syn_moments_b)");
            xz = P<T>(s.xz);
        } else {
            LCXCHECK(detail(h, 0, nullptr, P<T>(c.op_a), nullptr));      // solve(ry, rho)^T of the resident set 0 (:280)
            xz = P<T>(c.op_a);
        }
        // operand A = Y, padded to Mp columns, whole on the device (n_rows x Mp elements: small beside the output)
        DevTemps tmps;
        T* yd = nullptr;
        const int64_t rows_pad = round_up(n_rows, 64);
        LCXCHECK(tmps.get(&yd, sizeof(T) * rows_pad * Mp));
        {
            std::vector<T> tmp((size_t)rows_pad * Mp, (T)0);
            const T* src = reinterpret_cast<const T*>(y_host);
            for (int64_t r = 0; r < n_rows; ++r)
                for (int j = 0; j < h->M; ++j) tmp[r * Mp + j] = src[r * h->M + j];
            HIPCHECK(hipMemcpyAsync(yd, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
        }
        {
            const size_t bytes = (size_t)n_rows * (size_t)ld_out * sizeof(T);
            if (bytes >= ((size_t)8 << 20)) {
                const uintptr_t pg = (uintptr_t)2 << 20;
                const uintptr_t a0 = ((uintptr_t)out_host + pg - 1) & ~(pg - 1), a1 = ((uintptr_t)out_host + bytes) & ~(pg - 1);
                if (a1 > a0) (void)madvise((void*)a0, (size_t)(a1 - a0), MADV_HUGEPAGE);
            }
        }
        T* out = P<T>(out_host);
        const int64_t nblk = cdiv(n_rows, brows);
        double ksec = 0.0;
        for (int64_t k = 0; k <= nblk; ++k) {
            if (k < nblk) {
                const int b = (int)(k & 1);
                const int64_t r0 = k * brows, nr = (n_rows - r0) < brows ? (n_rows - r0) : brows;
                dim3 grid((unsigned)cdiv(V, 64), (unsigned)cdiv(nr, 64));
                HIPCHECK(hipEventRecord(c.t_a[b], h->stream));
                hipLaunchKernelGGL((cov_syrk_kernel<T, Mp, true>), grid, dim3(256), 0, h->stream, yd + r0 * Mp, xz, P<T>(c.std_dev), V, (int64_t)0, nr,
                                   (T)1, P<T>(c.dev[b]), ldo, P<T>(c.mean_dev), kind);
                KCHECK();
                HIPCHECK(hipEventRecord(c.t_b[b], h->stream));
                HIPCHECK(hipEventRecord(c.ev_k[b], h->stream));
                HIPCHECK(hipStreamWaitEvent(c.copy_stream, c.ev_k[b], 0));
                HIPCHECK(hipMemcpy2DAsync(c.pin[b], (size_t)V * sizeof(T), c.dev[b], (size_t)ldo * sizeof(T), (size_t)V * sizeof(T), (size_t)nr,
                                          hipMemcpyDeviceToHost, c.copy_stream));
                HIPCHECK(hipEventRecord(c.ev_c[b], c.copy_stream));
            }
            if (k >= 1) {
                const int b = (int)((k - 1) & 1);
                const int64_t nr = (n_rows - (k - 1) * brows) < brows ? (n_rows - (k - 1) * brows) : brows;
                HIPCHECK(hipEventSynchronize(c.ev_c[b]));
                {
                    float ms = 0.f;
                    HIPCHECK(hipEventElapsedTime(&ms, c.t_a[b], c.t_b[b]));
                    ksec += (double)ms * 1e-3;
                }
                place_rows(P<T>(c.pin[b]), V, out + (k - 1) * brows * ld_out, ld_out, nr, V);
            }
        }
        c.last_kernel_seconds = ksec;
        if (kernel_seconds) *kernel_seconds = ksec;
        return LCX_OK;
    }

    // invert (:431-438) of host rows: staged blocks, elementwise on the device
    static int invert_rows(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean_h, const void* std_h,
                           void* out_host, int64_t ld_out) {
        LCXCHECK(cov_stage(h, false, false));
        CovStage& c = *h->cov;
        const int64_t V = h->V, ldo = h->ldx, brows = c.block_rows;
        LCXCHECK(stage_theta(h, kind, mean_h, std_h));
        const T* x = reinterpret_cast<const T*>(x_host);
        T* out = P<T>(out_host);
        for (int64_t r0 = 0; r0 < n_rows; r0 += brows) {
            const int64_t nr = (n_rows - r0) < brows ? (n_rows - r0) : brows;
            HIPCHECK(hipMemcpy2DAsync(c.dev[0], (size_t)ldo * sizeof(T), x + r0 * ld, (size_t)ld * sizeof(T), (size_t)V * sizeof(T), (size_t)nr,
                                      hipMemcpyHostToDevice, h->stream));
            const int64_t total = nr * V;
            hipLaunchKernelGGL((invert_rows_kernel<T>), dim3((unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096)), dim3(256), 0, h->stream,
                               P<T>(c.dev[0]), nr, V, ldo, P<T>(c.mean_dev), P<T>(c.std_dev), kind, P<T>(c.dev[0]));
            KCHECK();
            HIPCHECK(hipMemcpy2DAsync(out + r0 * ld_out, (size_t)ld_out * sizeof(T), c.dev[0], (size_t)ldo * sizeof(T), (size_t)V * sizeof(T),
                                      (size_t)nr, hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
        }
        return LCX_OK;
    }

    // y[rows_pad][Mp] = xd[rows_pad][ldx] . W^T for a staged block of new rows (transform, :386-395).  Up to 128 padded
    // factors: the row-streaming kernel gemm_nt.  256: its register tile does not fit, so the block is transposed and runs
    // through the column-streaming kernel like the resident passes do.
    static int project_block(lcx_ctx* h, DevTemps& tmps, T* xd, int64_t rows_pad, T* yd, T** xt_io) {
        if constexpr (WIDE) {
            (void)tmps; (void)xt_io;
            return wide_gemm<false, false>(h, xd, h->ldx, P<T>(h->Wt[0]), Mp, nullptr, yd, Mp, rows_pad, Mp, h->ldx, 1, nullptr);
        } else if constexpr (CT <= 8) {
            (void)tmps; (void)xt_io;
            return launch_nt<T, CT>(h->stream, xd, h->ldx, rows_pad, P<T>(h->Wt[0]), yd, 1, 4, nullptr);
        } else {
            if (!*xt_io) LCXCHECK(tmps.get(xt_io, sizeof(T) * rows_pad * h->ldx));
            dim3 grid((unsigned)(h->ldx / 64), (unsigned)(rows_pad / 64));
            hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, h->stream, xd, h->ldx, *xt_io, rows_pad);
            KCHECK();
            return launch_tn<T, CT, Geo<T, CT>::TN_RT, false, false>(h->stream, *xt_io, rows_pad, h->ldx, rows_pad, P<T>(h->Wt[0]), nullptr, yd, 1,
                                                                      4, nullptr);
        }
    }

    static int project(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, void* out_host) {
        const int64_t blk = 8192;      // rows per staged block
        const int64_t rows_pad = round_up(n_rows < blk ? n_rows : blk, 64);
        T *xd = nullptr, *yd = nullptr, *xt_tmp = nullptr;
        DevTemps tmps;
        LCXCHECK(tmps.get(&xd, sizeof(T) * rows_pad * h->ldx));
        LCXCHECK(tmps.get(&yd, sizeof(T) * rows_pad * Mp));
        std::vector<T> tmp((size_t)rows_pad * Mp);
        T* out = P<T>(out_host);
        for (int64_t r0 = 0; r0 < n_rows; r0 += blk) {
            const int64_t nr = (n_rows - r0) < blk ? (n_rows - r0) : blk;
            HIPCHECK(hipMemsetAsync(xd, 0, sizeof(T) * rows_pad * h->ldx, h->stream));
            HIPCHECK(hipMemcpy2DAsync(xd, h->ldx * sizeof(T), reinterpret_cast<const T*>(x_host) + r0 * ld, ld * sizeof(T),
                                      h->V * sizeof(T), nr, hipMemcpyHostToDevice, h->stream));
            LCXCHECK(project_block(h, tmps, xd, rows_pad, yd, &xt_tmp));
            LCXCHECK(exchange(h, yd, rows_pad * Mp, DT));        // every rank projects the same rows: sum of the per-shard partials
            HIPCHECK(hipMemcpyAsync(tmp.data(), yd, sizeof(T) * rows_pad * Mp, hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            for (int64_t r = 0; r < nr; ++r)
                for (int j = 0; j < h->M; ++j) out[(r0 + r) * h->M + j] = tmp[r * Mp + j];
        }
        return LCX_OK;
    }

    // split mode (gemm_split_kernels.hpp) needs the panel-major copy, float32 and 32 / 64 / 128 padded factors
    static int split_supported(lcx_ctx* h) {
        if constexpr (!WIDE && split_capable<T, CT>()) {
            return h->panel ? 1 : 0;
        } else {
            (void)h;
            return 0;
        }
    }

    // Name of the kernel instantiation behind the two X-streaming passes, as rocprofv3 prints it (both
    // passes run the same function: X.B^T contracts over the rows of the transposed copy).
    static int kernel_name(lcx_ctx* h, int kind, char* buf, int64_t len) {
        if constexpr (WIDE) {
            if (kind == 2) buf[0] = 0;
            else snprintf(buf, (size_t)len, "lcx::gemm_wide_kernel<%s, %s, false>", sizeof(T) == 8 ? "double" : "float", kind == 0 ? "false" : "true");
            return LCX_OK;
        } else {
            return kernel_name_tuned(h, kind, buf, len);
        }
    }
    static int kernel_name_tuned(lcx_ctx* h, int kind, char* buf, int64_t len) {
        if (h->split) {
            if (kind == 2 && !h->merged_ok) { buf[0] = 0; return LCX_OK; }
            const int ct = kind == 2 ? 2 * CT : CT;
            snprintf(buf, (size_t)len, "lcx::gemm_split_kernel<%d, %d, 6, %s, true, false, 2, %d, %d>", ct, ct >= 4 ? 8 : 4, kind == 1 ? "true" : "false",
                     ct <= 4 ? 2 : 1, ct == 8 ? 2 : 1);
            return LCX_OK;
        }
        if (kind == 2) {
            if (!h->merged_ok) { buf[0] = 0; return LCX_OK; }
            if constexpr (CT <= 4)
                snprintf(buf, (size_t)len, h->panel ? "lcx::gemm_cr_kernel<%s, %d, %d, %d, %d, true, true>"
                                           : h->single_copy ? "lcx::gemm_cr_kernel<%s, %d, %d, %d, %d, false, false>" : "lcx::gemm_ct_kernel<%s, %d, %d, %d, %d, true, false>",
                         sizeof(T) == 8 ? "double" : "float", 2 * CT, CtShape<T, 2 * CT>::RT, ct_kw<T, 2 * CT>(h->panel), CtShape<T, 2 * CT>::U);
            return LCX_OK;
        }
        if (h->panel) {
            snprintf(buf, (size_t)len, kind == 0 ? "lcx::gemm_cr_kernel<%s, %d, %d, %d, %d, true, true>" : "lcx::gemm_ct_kernel<%s, %d, %d, %d, %d, true, true>",
                     sizeof(T) == 8 ? "double" : "float", CT, CtShape<T, CT>::RT, ct_kw<T, CT>(true), CtShape<T, CT>::U);
            return LCX_OK;
        }
        if (kind == 0 && h->single_copy) {
            snprintf(buf, (size_t)len, "lcx::gemm_cr_kernel<%s, %d, %d, %d, %d, false, false>", sizeof(T) == 8 ? "double" : "float", CT, CtShape<T, CT>::RT,
                     ct_kw<T, CT>(false), CtShape<T, CT>::U);
            return LCX_OK;
        }
        if (kind == 0 ? h->nt_ct : h->tn_ct)
            snprintf(buf, (size_t)len, "lcx::gemm_ct_kernel<%s, %d, %d, %d, %d, true, false>", sizeof(T) == 8 ? "double" : "float", CT,
                     CtShape<T, CT>::RT, ct_kw<T, CT>(false), CtShape<T, CT>::U);
        else if (h->f64_4x4)
            snprintf(buf, (size_t)len, "lcx::gemm_tn4_kernel<%d, %d, %d, 4, true, false>", CT, Geo<T, CT>::TN_RT, kind == 0 ? h->nt_KW : h->tn_KW);
        else
            snprintf(buf, (size_t)len, "lcx::gemm_tn_kernel<%s, %d, %d, %d, false, 4, false>", sizeof(T) == 8 ? "double" : "float", CT,
                     Geo<T, CT>::TN_RT, kind == 0 ? h->nt_KW : h->tn_KW);
        return LCX_OK;
    }

    // ---- preprocess on device (:397-429): stats + impute + standardise / tail squash, in place ----
    // on a ROW-MAJOR view X[Npad][ldx] of V columns: the resident shard itself, or - panel layout - a staged block of its columns
    // (every step of :397-429 is per column, so blocks of columns are preprocessed independently).
    // mean_io / std_io: host arrays of T (V entries: the caller offsets them to the view's first column); nobs_out: int64 (may be null);
    // xt_view: [ldx][Npad] to leave the transposed copy in when kind is 'empirical' (nullptr: a temporary for the sort)
    static int preprocess_resident(lcx_ctx* h, int kind, int has_missing, double sentinel, int fit, void* mean_io,
                                   void* std_io, int64_t* nobs_out, double* maxabs_out) {
        return preprocess_view(h, P<T>(h->X), h->V, h->ldx, kind, has_missing, sentinel, fit, mean_io, std_io, nobs_out, maxabs_out,
                               h->single_copy ? (T*)nullptr : P<T>(h->XT));
    }
    static int preprocess_view(lcx_ctx* h, T* X, const int64_t V, const int64_t ldx, int kind, int has_missing, double sentinel, int fit,
                               void* mean_io, void* std_io, int64_t* nobs_out, double* maxabs_out, T* xt_view) {
        const int64_t N = h->N;
        const int strips = (int)cdiv(V, 64);
        int RS = (int)cdiv(4 * h->n_cus, strips);
        if (RS > 64) RS = 64;
        if ((int64_t)RS * 16 > N) RS = (int)(N / 16 > 0 ? N / 16 : 1);
        if (RS < 1) RS = 1;
        double *nobs = nullptr, *imp = nullptr, *mean = nullptr, *stdv = nullptr, *ps = nullptr, *pn = nullptr, *bmax = nullptr;
        DevTemps tmps;
        LCXCHECK(tmps.get(&nobs, sizeof(double) * V));
        LCXCHECK(tmps.get(&imp, sizeof(double) * V));
        LCXCHECK(tmps.get(&mean, sizeof(double) * V));
        LCXCHECK(tmps.get(&stdv, sizeof(double) * V));
        LCXCHECK(tmps.get(&ps, sizeof(double) * V * RS));
        LCXCHECK(tmps.get(&pn, sizeof(double) * V * RS));
        LCXCHECK(tmps.get(&bmax, sizeof(double) * strips * RS));
        const dim3 grid((unsigned)strips, (unsigned)RS);
        const unsigned fgrid = (unsigned)cdiv(V, 256);
        const bool empirical = kind == PP_KIND_EMPIRICAL;        // (:424-426) imputation as usual, then ranks: no theta
        if (empirical) kind = PP_KIND_NONE;
        const bool need_stats = kind != PP_KIND_NONE;
        if (has_missing || (fit && need_stats)) {
            hipLaunchKernelGGL((pp_colsum_kernel<T>), grid, dim3(256), 0, h->stream, X, N, V, ldx, has_missing, (T)sentinel,
                               (const double*)nullptr, ps, pn);
            KCHECK();
            hipLaunchKernelGGL((pp_finalize_kernel<T>), dim3(fgrid), dim3(256), 0, h->stream, ps, pn, RS, V, (double)N, kind, 0, nobs, imp, stdv);
            KCHECK();
        }
        std::vector<double> tmp((size_t)V);
        if (need_stats) {
            if (fit) {
                HIPCHECK(hipMemcpyAsync(mean, imp, sizeof(double) * V, hipMemcpyDeviceToDevice, h->stream));
                hipLaunchKernelGGL((pp_colsum_kernel<T>), grid, dim3(256), 0, h->stream, X, N, V, ldx, has_missing, (T)sentinel,
                                   (const double*)mean, ps, (double*)nullptr);
                KCHECK();
                hipLaunchKernelGGL((pp_finalize_kernel<T>), dim3(fgrid), dim3(256), 0, h->stream, ps, (const double*)nullptr, RS, V, (double)N, kind,
                                   1, nobs, mean, stdv);
                KCHECK();
            } else {
                if (!mean_io || !std_io) return fail(LCX_ERR_ARG, "preprocess: theta required when fit == 0");
                const T* mh = reinterpret_cast<const T*>(mean_io);
                const T* sh = reinterpret_cast<const T*>(std_io);
                for (int64_t c = 0; c < V; ++c) tmp[c] = (double)mh[c];
                HIPCHECK(hipMemcpy(mean, tmp.data(), sizeof(double) * V, hipMemcpyHostToDevice));
                for (int64_t c = 0; c < V; ++c) tmp[c] = (double)sh[c];
                HIPCHECK(hipMemcpy(stdv, tmp.data(), sizeof(double) * V, hipMemcpyHostToDevice));
            }
        }
        if (need_stats || has_missing) {
            hipLaunchKernelGGL((pp_apply_kernel<T>), grid, dim3(256), 0, h->stream, X, N, V, ldx, has_missing, (T)sentinel, imp, mean, stdv,
                               kind, bmax);
            KCHECK();
        }
        if (empirical) {
            T* xt = xt_view;
            if (!xt) LCXCHECK(tmps.get(&xt, sizeof(T) * (size_t)h->Npad * ldx));      // the sort works on contiguous columns
            dim3 tg((unsigned)(ldx / 64), (unsigned)(h->Npad / 64));
            hipLaunchKernelGGL((transpose_kernel<T>), tg, dim3(256), 0, h->stream, X, ldx, xt, h->Npad);
            KCHECK();
            std::string err;
            if (empirical_columns<T>(X, ldx, xt, h->Npad, N, V, h->stream, &err) != 0) return fail(LCX_ERR_HIP, err);
        }
        HIPCHECK(hipStreamSynchronize(h->stream));
        if (fit && need_stats && mean_io && std_io) {
            T* mh = reinterpret_cast<T*>(mean_io);
            T* sh = reinterpret_cast<T*>(std_io);
            HIPCHECK(hipMemcpy(tmp.data(), mean, sizeof(double) * V, hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < V; ++c) mh[c] = (T)tmp[c];
            HIPCHECK(hipMemcpy(tmp.data(), stdv, sizeof(double) * V, hipMemcpyDeviceToHost));
            for (int64_t c = 0; c < V; ++c) sh[c] = (T)tmp[c];
        }
        if (nobs_out) {
            if (has_missing) {
                HIPCHECK(hipMemcpy(tmp.data(), nobs, sizeof(double) * V, hipMemcpyDeviceToHost));
                for (int64_t c = 0; c < V; ++c) nobs_out[c] = (int64_t)tmp[c];
            } else {
                for (int64_t c = 0; c < V; ++c) nobs_out[c] = N;
            }
        }
        if (maxabs_out) {
            *maxabs_out = 0.0;
            if (need_stats || has_missing) {
                std::vector<double> bm((size_t)strips * RS);
                HIPCHECK(hipMemcpy(bm.data(), bmax, sizeof(double) * bm.size(), hipMemcpyDeviceToHost));
                for (double v : bm) if (v > *maxabs_out) *maxabs_out = v;
            }
        }
        return LCX_OK;
    }

    // ---- panel layout: the shard is filled through a row-major staging block of columns -------------------------------------
    // fill(stage, ld, c0, nvalid) produces columns [c0, c0 + nvalid) of the shard (rows [0, N)) row-major in `stage` (leading dimension
    // ld, zeroed beforehand: that is the padding); the block is then scattered into its panels.  <= 2^28 staged elements.
    static int64_t panel_block_cols(const lcx_ctx* h) {
        int64_t w = (((int64_t)1 << 28) / h->Npad) / 64 * 64;
        const int forced = env_int("LCX_PANEL_BLOCK_COLS", 0);          // test hook: several blocks at small sizes
        if (forced > 0) w = (int64_t)forced / 64 * 64;
        if (w < 64) w = 64;
        return w > h->ldx ? h->ldx : w;
    }
    template <typename F> static int panel_fill(lcx_ctx* h, F fill) {
        const int64_t W = panel_block_cols(h);
        DevTemps tmps;
        T* stage = nullptr;
        LCXCHECK(tmps.get(&stage, sizeof(T) * (size_t)h->Npad * W));
        for (int64_t c0 = 0; c0 < h->ldx; c0 += W) {
            const int64_t wp = (h->ldx - c0) < W ? (h->ldx - c0) : W;
            const int64_t wv = h->V - c0 < 0 ? 0 : (h->V - c0 < wp ? h->V - c0 : wp);
            HIPCHECK(hipMemsetAsync(stage, 0, sizeof(T) * (size_t)h->Npad * W, h->stream));
            if (wv > 0) LCXCHECK(fill(stage, W, c0, wv));
            hipLaunchKernelGGL((panel_block_kernel<T, true>), dim3(4096), dim3(256), 0, h->stream, stage, W, P<T>(h->X), h->Npad * PanelW<T>::v,
                               h->Npad, c0, wp);
            KCHECK();
        }
        HIPCHECK(hipStreamSynchronize(h->stream));
        return LCX_OK;
    }
    static int upload_x(lcx_ctx* h, const void* x, int64_t ld) {
        if (!h->panel) {
            HIPCHECK(hipMemcpy2DAsync(h->X, h->ldx * sizeof(T), x, ld * sizeof(T), h->V * sizeof(T), h->N, hipMemcpyHostToDevice, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            return make_xt(h);
        }
        return panel_fill(h, [&](T* stage, int64_t lds, int64_t c0, int64_t wv) -> int {
            HIPCHECK(hipMemcpy2DAsync(stage, lds * sizeof(T), reinterpret_cast<const T*>(x) + c0, ld * sizeof(T), wv * sizeof(T), h->N,
                                      hipMemcpyHostToDevice, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));       // (the next block's memset must not overtake a pageable-memory copy)
            return LCX_OK;
        });
    }
    static int download_x(lcx_ctx* h, void* x, int64_t ld) {
        if (!h->panel) {
            HIPCHECK(hipMemcpy2DAsync(x, ld * sizeof(T), h->X, h->ldx * sizeof(T), h->V * sizeof(T), h->N, hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            return LCX_OK;
        }
        const int64_t W = panel_block_cols(h);
        DevTemps tmps;
        T* stage = nullptr;
        LCXCHECK(tmps.get(&stage, sizeof(T) * (size_t)h->Npad * W));
        for (int64_t c0 = 0; c0 < h->V; c0 += W) {
            const int64_t wp = (h->ldx - c0) < W ? (h->ldx - c0) : W;
            const int64_t wv = h->V - c0 < wp ? h->V - c0 : wp;
            hipLaunchKernelGGL((panel_block_kernel<T, false>), dim3(4096), dim3(256), 0, h->stream, stage, W, P<T>(h->X), h->Npad * PanelW<T>::v,
                               h->Npad, c0, wp);
            KCHECK();
            HIPCHECK(hipMemcpy2DAsync(reinterpret_cast<T*>(x) + c0, ld * sizeof(T), stage, W * sizeof(T), wv * sizeof(T), h->N, hipMemcpyDeviceToHost,
                                      h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
        }
        return LCX_OK;
    }

    static int upload_preprocess(lcx_ctx* h, const void* x, int64_t ld, int kind, int has_missing, double sentinel, int fit,
                                 void* mean_io, void* std_io, int64_t* nobs_out, double* maxabs_out) {
        if (!h->panel) {
            HIPCHECK(hipMemcpy2DAsync(h->X, h->ldx * sizeof(T), x, ld * sizeof(T), h->V * sizeof(T), h->N, hipMemcpyHostToDevice, h->stream));
            LCXCHECK(preprocess_resident(h, kind, has_missing, sentinel, fit, mean_io, std_io, nobs_out, maxabs_out));
            return make_xt(h);
        }
        double mx_all = 0.0;
        LCXCHECK(panel_fill(h, [&](T* stage, int64_t lds, int64_t c0, int64_t wv) -> int {
            HIPCHECK(hipMemcpy2DAsync(stage, lds * sizeof(T), reinterpret_cast<const T*>(x) + c0, ld * sizeof(T), wv * sizeof(T), h->N,
                                      hipMemcpyHostToDevice, h->stream));
            double mx = 0.0;
            LCXCHECK(preprocess_view(h, stage, wv, lds, kind, has_missing, sentinel, fit, mean_io ? (void*)(reinterpret_cast<T*>(mean_io) + c0) : nullptr,
                                     std_io ? (void*)(reinterpret_cast<T*>(std_io) + c0) : nullptr, nobs_out ? nobs_out + c0 : nullptr, &mx,
                                     (T*)nullptr));
            if (mx > mx_all) mx_all = mx;
            return LCX_OK;
        }));
        if (maxabs_out) *maxabs_out = mx_all;
        return LCX_OK;
    }

    // transform (:386-395) of raw rows: standardise with theta on the device, then x~ . ws^T
    static int project_raw(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean_h,
                           const void* std_h, void* out_host) {
        const int64_t blk = 8192;
        const int64_t rows_pad = round_up(n_rows < blk ? n_rows : blk, 64);
        T *xd = nullptr, *yd = nullptr, *xt_tmp = nullptr;
        double *mean = nullptr, *stdv = nullptr, *bmax = nullptr;
        const int strips = (int)cdiv(h->V, 64);
        const int RS = 8;
        DevTemps tmps;
        LCXCHECK(tmps.get(&xd, sizeof(T) * rows_pad * h->ldx));
        LCXCHECK(tmps.get(&yd, sizeof(T) * rows_pad * Mp));
        LCXCHECK(tmps.get(&mean, sizeof(double) * h->V));
        LCXCHECK(tmps.get(&stdv, sizeof(double) * h->V));
        LCXCHECK(tmps.get(&bmax, sizeof(double) * strips * RS));
        if (kind != PP_KIND_NONE) {
            std::vector<double> tmp((size_t)h->V);
            for (int64_t c = 0; c < h->V; ++c) tmp[c] = (double)reinterpret_cast<const T*>(mean_h)[c];
            HIPCHECK(hipMemcpy(mean, tmp.data(), sizeof(double) * h->V, hipMemcpyHostToDevice));
            for (int64_t c = 0; c < h->V; ++c) tmp[c] = (double)reinterpret_cast<const T*>(std_h)[c];
            HIPCHECK(hipMemcpy(stdv, tmp.data(), sizeof(double) * h->V, hipMemcpyHostToDevice));
        }
        std::vector<T> tmp((size_t)rows_pad * Mp);
        T* out = P<T>(out_host);
        for (int64_t r0 = 0; r0 < n_rows; r0 += blk) {
            const int64_t nr = (n_rows - r0) < blk ? (n_rows - r0) : blk;
            HIPCHECK(hipMemsetAsync(xd, 0, sizeof(T) * rows_pad * h->ldx, h->stream));
            HIPCHECK(hipMemcpy2DAsync(xd, h->ldx * sizeof(T), reinterpret_cast<const T*>(x_host) + r0 * ld, ld * sizeof(T),
                                      h->V * sizeof(T), nr, hipMemcpyHostToDevice, h->stream));
            if (kind != PP_KIND_NONE) {
                hipLaunchKernelGGL((pp_apply_kernel<T>), dim3((unsigned)strips, RS), dim3(256), 0, h->stream, xd, nr, h->V, h->ldx, 0, (T)0,
                                   (const double*)nullptr, mean, stdv, kind, bmax);
                KCHECK();
            }
            LCXCHECK(project_block(h, tmps, xd, rows_pad, yd, &xt_tmp));
            LCXCHECK(exchange(h, yd, rows_pad * Mp, DT));        // every rank projects the same rows: sum of the per-shard partials
            HIPCHECK(hipMemcpyAsync(tmp.data(), yd, sizeof(T) * rows_pad * Mp, hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            for (int64_t r = 0; r < nr; ++r)
                for (int j = 0; j < h->M; ++j) out[(r0 + r) * h->M + j] = tmp[r * Mp + j];
        }
        return LCX_OK;
    }

    static int generate(lcx_ctx* h, uint64_t seed, int kind, int n_groups, int64_t col_offset) {
        if (h->panel)          // the generator is keyed by (seed, row, global column): block by block gives the same matrix
            return panel_fill(h, [&](T* stage, int64_t lds, int64_t c0, int64_t wv) -> int {
                hipLaunchKernelGGL((generate_kernel<T>), dim3(4096), dim3(256), 0, h->stream, stage, h->N, wv, lds, seed, kind,
                                   n_groups < 1 ? 1 : n_groups, col_offset + c0);
                KCHECK();
                return preprocess_view(h, stage, wv, lds, PP_KIND_STANDARD, 0, 0.0, 1, nullptr, nullptr, nullptr, nullptr, (T*)nullptr);
            });
        hipLaunchKernelGGL((generate_kernel<T>), dim3(4096), dim3(256), 0, h->stream, P<T>(h->X), h->N, h->V, h->ldx,
                           seed, kind, n_groups < 1 ? 1 : n_groups, col_offset);
        KCHECK();
        // standardise like preprocess 'standard' (:409-415)
        LCXCHECK(preprocess_resident(h, PP_KIND_STANDARD, 0, 0.0, 1, nullptr, nullptr, nullptr, nullptr));
        return make_xt(h);
    }
};

// runtime (dtype, CT) -> Impl<T, CT>::fn(args...)
#define DISPATCH(h, fn, ...)                                                        \
    do {                                                                            \
        if ((h)->dtype == LCX_F32) {                                                \
            switch ((h)->CT) {                                                      \
                case 1: return Impl<float, 1>::fn(__VA_ARGS__);                     \
                case 2: return Impl<float, 2>::fn(__VA_ARGS__);                     \
                case 4: return Impl<float, 4>::fn(__VA_ARGS__);                     \
                case 8: return Impl<float, 8>::fn(__VA_ARGS__);                     \
                case 16: return Impl<float, 16>::fn(__VA_ARGS__);                   \
                case 32: return Impl<float, 32>::fn(__VA_ARGS__);                   \
                case 64: return Impl<float, 64>::fn(__VA_ARGS__);                   \
            }                                                                       \
        } else {                                                                    \
            switch ((h)->CT) {                                                      \
                case 1: return Impl<double, 1>::fn(__VA_ARGS__);                    \
                case 2: return Impl<double, 2>::fn(__VA_ARGS__);                    \
                case 4: return Impl<double, 4>::fn(__VA_ARGS__);                    \
                case 8: return Impl<double, 8>::fn(__VA_ARGS__);                    \
                case 16: return Impl<double, 16>::fn(__VA_ARGS__);                  \
                case 32: return Impl<double, 32>::fn(__VA_ARGS__);                  \
                case 64: return Impl<double, 64>::fn(__VA_ARGS__);                  \
            }                                                                       \
        }                                                                           \
        return fail(LCX_ERR_ARG, "unsupported n_hidden padding");                   \
    } while (0)

static int ct_for(int m) {
    if (m <= 16) return 1;
    if (m <= 32) return 2;
    if (m <= 64) return 4;
    if (m <= 128) return 8;
    if (m <= 256) return 16;
    if (m <= 512) return 32;           // the wide path (gemm_wide): 512 / 1024 padded factors
    if (m <= 1024) return 64;
    return 0;
}

// (the read of the per-thread last error drops whatever an earlier, unrelated HIP call of this thread left behind: the
// launch checks of this call must only see this call's errors)
#define NEED(h)                                            \
    if (!(h)) return fail(LCX_ERR_ARG, "null handle");     \
    HIPCHECK(hipSetDevice((h)->device));                   \
    (void)hipGetLastError();

// Every entry point that changes the fit state abandons a speculation of lcx_iterate (its kernels run to completion on
// buffers only a trial owns; what they leave behind in the shared exchange buffer is remembered in spec_dirty).
static inline void cancel_speculation(lcx_ctx* h) {
    if (h->spec_pending) {
        h->spec_pending = false;
        h->early_grad = h->grad_ready = false;
        h->spec_dirty = true;
        h->have_direction = false;
        h->w1_ready = h->y1_ready = h->yk_ready = false;
    }
}
#define NEED_MUT(h) NEED(h); cancel_speculation(h); (h)->early_grad = (h)->grad_ready = (h)->yk_ready = false

#define WHICH_OK(w) if ((w) < 0 || (w) > 1) return fail(LCX_ERR_ARG, "which must be 0 or 1")

// Wait until the pinned mirror of a set carries the last publication enqueued for it.
// zero-filled device memory
static int dev_alloc(void** p, size_t bytes, hipStream_t st) {
    HIPCHECK(hipMalloc(p, bytes ? bytes : 16));
    HIPCHECK(hipMemsetAsync(*p, 0, bytes ? bytes : 16, st));
    return LCX_OK;
}

static int wait_published(lcx_ctx* h, MomentSet& s) {
    if (s.seq_expect == 0) {                 // nothing published yet: plain copy
        HIPCHECK(hipMemcpyAsync(s.hst, s.st, sizeof(SetState), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        return LCX_OK;
    }
    // The wait is bounded (LCX_WAIT_TIMEOUT_MS, default 120 s): a rank whose publication never arrives fails with
    // what it expected and what it saw instead of hanging the whole job in the next collective.
    static const double limit_s = []() {
        const char* e = getenv("LCX_WAIT_TIMEOUT_MS");
        return (e && *e) ? atof(e) * 1e-3 : 120.0;
    }();
    volatile unsigned int* p = &s.hst->seq;
    int spins = 0;
    timespec t0{0, 0};
    bool have_t0 = false;
    while (__atomic_load_n(p, __ATOMIC_ACQUIRE) != s.seq_expect) {
        if (++spins >= 4096) {
            spins = 0;
            hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == s.seq_expect) break;
                return fail(LCX_ERR_STATE, "stream drained but the state mirror was not published: expected seq " +
                                               std::to_string(s.seq_expect) + ", mirror has " + std::to_string(*p));
            }
            if (q != hipErrorNotReady) HIPCHECK(q);
            timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            if (!have_t0) { t0 = now; have_t0 = true; }
            const double waited = (double)(now.tv_sec - t0.tv_sec) + 1e-9 * (double)(now.tv_nsec - t0.tv_nsec);
            if (waited > limit_s)
                return fail(LCX_ERR_STATE, "state mirror not published after " + std::to_string(waited) + " s: expected seq " +
                                               std::to_string(s.seq_expect) + ", mirror has " + std::to_string(*p) +
                                               ", last enqueued seq " + std::to_string(h->seq_next) +
                                               ", stream still busy (hipErrorNotReady), world " + std::to_string(h->world));
            sched_yield();                   // several ranks of one box may share few cores (tests: two ranks + gloo threads)
        }
        __builtin_ia32_pause();
    }
    return LCX_OK;
}

// the launch geometry of a handle (lcx_create): lives with the kernels it sizes, in lcx_levels.hip
__attribute__((visibility("hidden"))) int lcx_engine_geometry(lcx_ctx* h);
