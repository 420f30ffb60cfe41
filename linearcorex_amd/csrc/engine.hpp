// engine.hpp - the engine behind the C ABI declared in include/lcx.h, shared by its translation units.
//
// Owns the device state of one n_variables shard of a Linear CorEx fit (X, W and two moment sets, all resident in HBM) and
// enqueues the kernels of each dependency level of the reference's _calculate_moments_ns / _update_ns (linearcorex.py:236-334)
// on one HIP stream.  No torch types, no callbacks.  This header holds the context, the launch helpers and the class `Impl<T, CT>` -
// the typed implementation of every level: its member DECLARATIONS are the index of the engine; the definitions live in three seam
// headers included at the end (impl_levels.hpp, impl_data.hpp, impl_outputs.hpp) and are instantiated by whichever translation
// unit's entry points call them:
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
constexpr int LCX_T_KINDS = 7;          // timing sites: 0-2 the X passes, 3-6 the exchange steps (lcx_ctx::t_launch)
constexpr int LCX_T_AR_Y = 3, LCX_T_AR_DIR = 4, LCX_T_AR_S = 5, LCX_T_AR_SMALL = 6;

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
    // Several ranks: whether a shard HAS a merged form depends on its own width (slot count of the 2 Mp-column launch), but taking it
    // changes the sequence of collectives (a Bj all-reduce in front of the pass, one [Y' | W'.W'^T | Y_g] all-reduce behind it): the
    // ranks agree once per transport - all of them, or none (-1: not asked yet)
    int merged_agreed;
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
    int t_every;                // HIP-event timing samples every t_every-th launch of a site (an event pair costs ~5 us of stream time)
    int t_count[LCX_T_KINDS];
    std::vector<TimingPair> pending;
    std::vector<TimingPair> pool;
    // sites: kind 0 = X.B^T, 1 = X^T.Y, 2 = the merged X.[grad | ws+update]^T pass (2 Mp columns); the exchange steps (events on the
    // stream that carries the collective): 3 = the Y-buffer all-reduce behind X.W^T (lcx_moments_a, :247 -> :259), 4 = the one of the
    // direction (lcx_update_b: [Y_g | Bj], merged form [Y' | W'.W'^T | Y_g]), 5 = the scalar buffer (TC sums, tangent, H: :294, :301-305),
    // 6 = the small ones (Bj in front of the merged pass, W'.W'^T of a trial taken by linearity, a restored H)
    int64_t t_launch[LCX_T_KINDS];
    int64_t t_pass[LCX_T_KINDS];   // every launch of the site issued while timing is on (sampled or not)
    double t_ms[LCX_T_KINDS];
    double t_max[LCX_T_KINDS];     // longest timed launch per kind; X passes below a fifth of it were skipped by their flag
    int64_t t_skipped[LCX_T_KINDS];
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
    // LCX_Y_PIPELINE=signal[:n[:poll]]: ONE pass launch that signals every row chunk of its partial tiles through a signal word
    // (gemm_kernels.hpp, ChunkSig); the second stream waits on the word (hipStreamWaitValue32, or poll_signal_kernel with ":poll" /
    // where the runtime has no wait-value), sums the chunk's slots and all-reduces it while the pass goes on.  Wave-split kernels only
    bool ypipe_signal, ypipe_poll;
    unsigned int *sig_counters;         // [SIG_MAX_CHUNKS + 1]: chunk tickets, poll error word
    unsigned int* sig_flag[SIG_MAX_CHUNKS];
    unsigned int sig_epoch;
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
// the second stream of the pipelined Y exchange (LCX_Y_PIPELINE) and its events, created on first use
static int ypipe_streams(lcx_ctx* h) {
    if (h->comm_stream) return LCX_OK;
    HIPCHECK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    for (auto& e : h->ypipe_ev) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return LCX_OK;
}
// the tickets and signal words of the in-launch chunk signalling, created on first use.  A signal word is its own 8-byte allocation of
// signal memory (what hipStreamWaitValue32 may wait on); where the runtime refuses that, plain device memory and the polling kernel
static int ypipe_signals(lcx_ctx* h) {
    if (h->sig_counters) return LCX_OK;
    const size_t words = SIG_MAX_CHUNKS + 1;
    HIPCHECK(hipMalloc((void**)&h->sig_counters, words * sizeof(unsigned int)));
    HIPCHECK(hipMemsetAsync(h->sig_counters, 0, words * sizeof(unsigned int), h->stream));
    int can_wait = 0;
    if (hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, h->device) != hipSuccess) can_wait = 0;
    (void)hipGetLastError();
    for (int c = 0; c < SIG_MAX_CHUNKS; ++c) {
        void* p = nullptr;
        if (!h->ypipe_poll && can_wait && hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess) p = nullptr;
        (void)hipGetLastError();
        if (!p) {
            h->ypipe_poll = true;                 // no signal memory: every chunk polls (mixing the two would be legal but pointless)
            HIPCHECK(hipMalloc(&p, 8));
        }
        h->sig_flag[c] = (unsigned int*)p;
        hipLaunchKernelGGL(init_signal_kernel, dim3(1), dim3(1), 0, h->stream, h->sig_flag[c], 0u);
        KCHECK();
    }
    h->sig_epoch = 0;
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}
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
static int timing_begin_on(lcx_ctx* h, hipStream_t st, int kind, TimingPair* tp) {
    tp->kind = -1;
    if (!h || !h->timing || kind < 0) return LCX_OK;
    h->t_pass[kind] += 1;
    if (h->t_every > 1 && (h->t_count[kind]++ % h->t_every) != 0) return LCX_OK;
    if (h->pool.empty()) {
        HIPCHECK(hipEventCreate(&tp->a));
        HIPCHECK(hipEventCreate(&tp->b));
    } else {
        *tp = h->pool.back();
        h->pool.pop_back();
    }
    tp->kind = kind;
    HIPCHECK(hipEventRecord(tp->a, st));
    return LCX_OK;
}
static int timing_end_on(lcx_ctx* h, hipStream_t st, int kind, TimingPair* tp) {
    if (!h || !h->timing || kind < 0 || tp->kind < 0) return LCX_OK;
    HIPCHECK(hipEventRecord(tp->b, st));
    h->pending.push_back(*tp);
    return LCX_OK;
}
static int timing_begin(lcx_ctx* h, int kind, TimingPair* tp) { return timing_begin_on(h, h ? h->stream : nullptr, kind, tp); }
static int timing_end(lcx_ctx* h, int kind, TimingPair* tp) { return timing_end_on(h, h ? h->stream : nullptr, kind, tp); }
// an exchange step with its timing site: the event pair brackets the collective on the stream that carries it (RCCL: the all-reduce
// kernel and whatever it waits for - a slower rank shows up here; a host-blocking hook: the device-side gap it leaves)
static int exchange_site_on(lcx_ctx* h, hipStream_t st, int site, void* buf, int64_t count, int dtype) {
    if (!h->exchange || h->tr.kind == 0 || count <= 0) return LCX_OK;
    TimingPair tp;
    LCXCHECK(timing_begin_on(h, st, site, &tp));
    LCXCHECK(exchange_on(h, st, buf, count, dtype));
    return timing_end_on(h, st, site, &tp);
}
static int exchange_site(lcx_ctx* h, int site, void* buf, int64_t count, int dtype) { return exchange_site_on(h, h->stream, site, buf, count, dtype); }
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
        if (tp.kind < 3 && dur[k] < 0.2 * h->t_max[tp.kind]) {
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

    // ---- definitions: impl_levels.hpp (launch geometry, X passes, levels, lcx_iterate, synergistic branch, readback) ----
    template <bool TRANS_A, bool SCALE>
    static int wide_gemm(lcx_ctx* h, const T* A, int64_t lda, const T* B, int64_t ldb, const T* scale, T* C, int64_t ldc, int64_t M,
                         int64_t N, int64_t K, int S, const int* skip);

    template <typename F> static int blocks_per_cu(F* f, int threads, size_t lds);
    static int env_int(const char* name, int dflt);
    static int single_round_split(int64_t tiles, int64_t capacity_blocks, int64_t kunits, int kw, int cap);

    static int geometry(lcx_ctx* h);
    static int geometry_tuned(lcx_ctx* h);

    // the X passes (X.B^T: nt_*, X^T.Y: tn_big), their slot reductions, the m x m Grams, the pipelined exchange
    static int gram(lcx_ctx* h, const T* A, int64_t K, const T* scale, int S, const int* skip, T* dst);

    static int nt_pass(lcx_ctx* h, const T* B, const int* skip, T* dst, int64_t r0 = 0, int64_t rows = -1);
    static int nt_reduce(lcx_ctx* h, const int* skip, T* also, int64_t e0, int64_t n, hipStream_t st = nullptr);
    static int nt_big(lcx_ctx* h, const void* Bv, const int* skip, bool with_bj = false, T* also = nullptr);

    static int ypipe_init(lcx_ctx* h);
    static int y_pass_pipelined(lcx_ctx* h, const T* w, bool with_bj);
    static int make_xt(lcx_ctx* h);
    static int tn_big(lcx_ctx* h, const int* skip);

    static int gram_w(lcx_ctx* h, const T* w);

    // _calculate_moments_ns (:236-275) by dependency level, the linear trial mode
    static int moments_a(lcx_ctx* h, int which);

    static int small(lcx_ctx* h, int which, double eps, int quick, const T* ysrc);

    static int gram_pair(lcx_ctx* h, const T* w, const T* y);
    static int gram_pair_tuned(lcx_ctx* h, const T* w, const T* y);

    static int epilogue(lcx_ctx* h, int which, double eps, bool linear, double eta);

    static int moments_b(lcx_ctx* h, int which, double eps, int quick);

    static int trial_linear_a(lcx_ctx* h, double eta);
    static int trial_linear_b(lcx_ctx* h, double eps, double eta);

    static int moments_c(lcx_ctx* h, int which);

    // _update_ns (:290-334): direction, trials, the whole iteration with its line search (lcx_iterate)
    static int update_a(lcx_ctx* h);

    static int launch_grad(lcx_ctx* h, int which);
    static int update_b(lcx_ctx* h, double eps);
    static int agree_on_merged(lcx_ctx* h);
    static bool use_merged(const lcx_ctx* h);
    static int update_grid(const lcx_ctx* h);

    static int update_c(lcx_ctx* h, double eps);

    static int make_trial(lcx_ctx* h, double eta);

    static int trial_by_linearity(lcx_ctx* h, double eta);

    static int evaluate_trial(lcx_ctx* h, double eps);
    static int direction_and_trial(lcx_ctx* h, double eps);
    static int iterate(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out);
    static int iterate_body(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out);

    // stage change, initial scale, factor sort, detail moments (:129-133, :117, :160-163, :277-287)
    static int rescale(lcx_ctx* h, double e0, double e1);

    static int init_scale(lcx_ctx* h);

    static int permute(lcx_ctx* h, const int32_t* order);

    static int detail(lcx_ctx* h, int which, T* mi_o, T* xz_o, T* x2y_o);

    // synergistic branch (:336-384)
    static int syn_alloc(lcx_ctx* h);
    static int syn_moments_b(lcx_ctx* h, int which, double yscale);
    static int syn_moments_c(lcx_ctx* h, int which);
    static int syn_update_a(lcx_ctx* h);
    static int syn_update_b(lcx_ctx* h, double eta);
    // ---- definitions: impl_outputs.hpp (get_covariance :443-455, predict :440-441, invert :431-438) ----
    static int cov_stage(lcx_ctx* h, bool need_op_a, bool need_op_b);
    static void place_rows(const T* src, int64_t src_ld, T* dst, int64_t dst_ld, int64_t rows, int64_t cols);
    static int covariance_blocks(lcx_ctx* h, bool syn, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out_host,
                                 int64_t ld_out, double* kernel_seconds);
    static int covariance_syn(lcx_ctx* h, const void* std_host, int64_t row0, int64_t nrows, void* out_host);

    // ---- impl_levels.hpp: readback / upload of weights and moments ----
    static int fetch_mv(lcx_ctx* h, const T* dev, T* host, bool as_m_by_v);
    static int fetch_v(lcx_ctx* h, const T* dev, T* host);
    static int fetch_small(lcx_ctx* h, const double* dev, int rows, int cols, T* host);

    static int get_moment(lcx_ctx* h, int which, int key, double eps, void* out);

    static int set_moment(lcx_ctx* h, int which, int key, const void* in);

    static int set_ws(lcx_ctx* h, const void* w_host);

    // ---- impl_outputs.hpp ----
    static int covariance_full(lcx_ctx* h, int syn, double eps, const void* std_host, void* out_host, int64_t ld_out, double* ksec);
    static int covariance(lcx_ctx* h, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out_host);

    static int stage_theta(lcx_ctx* h, int kind, const void* mean_h, const void* std_h);

    static int predict(lcx_ctx* h, const void* y_host, int64_t n_rows, int syn, const void* xz_host, int kind, const void* mean_h,
                       const void* std_h, void* out_host, int64_t ld_out, double* kernel_seconds);

    static int invert_rows(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean_h, const void* std_h,
                           void* out_host, int64_t ld_out);

    // ---- definitions: impl_data.hpp (transform of new rows :386-395; upload + preprocess :397-429, generator, download) ----
    static int project_block(lcx_ctx* h, DevTemps& tmps, T* xd, int64_t rows_pad, T* yd, T** xt_io);

    static int project(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, void* out_host);

    // ---- impl_levels.hpp: what the shard supports, which kernels it runs ----
    static int split_supported(lcx_ctx* h);

    static int kernel_name(lcx_ctx* h, int kind, char* buf, int64_t len);
    static int kernel_name_tuned(lcx_ctx* h, int kind, char* buf, int64_t len);

    // ---- impl_data.hpp ----
    static int preprocess_resident(lcx_ctx* h, int kind, int has_missing, double sentinel, int fit, void* mean_io,
                                   void* std_io, int64_t* nobs_out, double* maxabs_out);
    static int preprocess_view(lcx_ctx* h, T* X, const int64_t V, const int64_t ldx, int kind, int has_missing, double sentinel, int fit,
                               void* mean_io, void* std_io, int64_t* nobs_out, double* maxabs_out, T* xt_view);

    static int64_t panel_block_cols(const lcx_ctx* h);
    template <typename F> static int panel_fill(lcx_ctx* h, F fill);
    static int upload_x(lcx_ctx* h, const void* x, int64_t ld);
    static int download_x(lcx_ctx* h, void* x, int64_t ld);

    static int upload_preprocess(lcx_ctx* h, const void* x, int64_t ld, int kind, int has_missing, double sentinel, int fit,
                                 void* mean_io, void* std_io, int64_t* nobs_out, double* maxabs_out);

    static int project_raw(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean_h,
                           const void* std_h, void* out_host);

    static int generate(lcx_ctx* h, uint64_t seed, int kind, int n_groups, int64_t col_offset);
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

// the member definitions of Impl<T, CT>, along the seams of the C ABI
#include "impl_levels.hpp"
#include "impl_data.hpp"
#include "impl_outputs.hpp"
