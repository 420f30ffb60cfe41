/*
 * lcx.h - C ABI of the MI355X-native Linear CorEx fit engine (liblcx_hip.so).
 *
 * This is the drop-in boundary for the reference's accelerator seam.  The reference
 * (linearcorex/linearcorex.py) switches on `self.gpu` and calls the un-vendored `cudamat`
 * package at six sites:
 *     cm.cublas_init()                       linearcorex.py:85-86
 *     cm.CUDAMatrix(x)   (upload X once)     linearcorex.py:427-428
 *     cm.dot(x, u.T) / cm.dot(x.T, y)        linearcorex.py:199-208   (_sig)
 *                                            linearcorex.py:217-224   (_norm)
 *                                            linearcorex.py:240-257   (_calculate_moments_ns)
 * i.e. its seam is "one SGEMM, then back to the host".  Here the seam moves up to whole
 * dependency levels of the algorithm so that X, W and every moment stay resident in HBM and the
 * host only reads a handful of scalars for control flow.  Each entry point below names the
 * reference lines it replaces.
 *
 * Conventions
 *   - every function returns an lcx_status (0 = OK); lcx_last_error() gives the message;
 *   - plain pointers and sizes only; host buffers are borrowed for the duration of the call;
 *   - device memory is owned by the handle, except the two exchange buffers which a multi-GPU
 *     caller may bind to its own allocations (lcx_bind_exchange) so that it can all-reduce them
 *     (RCCL through torch.distributed in linearcorex_amd/comm.py);
 *   - one caller per handle at a time (the reference is single-threaded too);
 *   - "which" selects a moment set: 0 = current solution (self.ws / self.moments),
 *     1 = line-search trial (w_update / m_update, linearcorex.py:320-321).
 *   - host matrices are C-order (row-major) in the reference's orientation: W and the "m by nv"
 *     moments are (n_hidden, nv_local).
 *
 * Multi-GPU: n_variables is sharded; every handle owns nv_local columns.  The quantities that
 * need a sum over all variables are written to the exchange buffers between the *_a/_b/_c
 * halves of a level; with one GPU the halves are simply called back to back.
 */
#ifndef LCX_H
#define LCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lcx_ctx lcx_ctx;

typedef enum {
    LCX_OK = 0,
    LCX_ERR_ARG = 1,        /* bad argument                                  */
    LCX_ERR_HIP = 2,        /* a HIP runtime call failed (see lcx_last_error) */
    LCX_ERR_NO_DEVICE = 3,  /* no usable gfx950 device                        */
    LCX_ERR_STATE = 4,      /* call sequence violated                         */
    LCX_ERR_COMM = 5        /* RCCL / the exchange hook failed                */
} lcx_status;

typedef enum { LCX_F32 = 0, LCX_F64 = 1 } lcx_dtype;

/* scalar slots returned by lcx_read_state() */
enum {
    LCX_S_TC = 0,        /* m["TC"]                    linearcorex.py:272-274 */
    LCX_S_MAX_UJ = 1,    /* max_j m["uj"]              linearcorex.py:250     */
    LCX_S_INVALID = 2,   /* 1.0 if quick and max uj>=1 (the `False` sentinel, :251) */
    LCX_S_TANGENT = 3,   /* update_tangent             linearcorex.py:305     */
    LCX_S_SUM_LOG_RJ = 4,/* sum_j log(1-uj)                                   */
    LCX_S_COUNT = 8
};

/* keys for lcx_get_moment(); shapes are the reference's (linearcorex.py:249-287) */
typedef enum {
    LCX_M_UJ = 0,          /* (m,)        */
    LCX_M_RHO = 1,         /* (m, nv)     */
    LCX_M_RY = 2,          /* (m, m)      */
    LCX_M_INVRHO = 3,      /* (m, nv)     */
    LCX_M_RHOINVRHO = 4,   /* (m, nv)     */
    LCX_M_QIJ = 5,         /* (m, nv)     */
    LCX_M_SI = 6,          /* (nv,)       */
    LCX_M_QISI2 = 7,       /* (nv,)       */
    LCX_M_MI = 8,          /* (m, nv)     */
    LCX_M_XIZJ = 9,        /* (nv, m)     */
    LCX_M_XI2_GIVEN_Y = 10,/* (nv,)       */
    LCX_M_GRAD = 11,       /* (m, nv)  last gradient      (:296-300) */
    LCX_M_UPDATE = 12,     /* (m, nv)  last update        (:303)     */
    LCX_M_SIG_GRAD = 13,   /* (m, nv)  last _sig(grad)    (:301)     */
    LCX_M_H = 14,          /* (m, m)   last H             (:294-295) */
    LCX_M_Y = 15,          /* (n_samples, m)  X.W^T of the set (:247); all-reduced if sharded */
    /* synergistic branch (_calculate_moments_syn, :336-373) */
    LCX_M_SYN_XIZJ = 16,   /* (nv, m)  X_i Z_j = solve(cy, X_i Y_j^T)^T   (:367) */
    LCX_M_SYN_X2Y = 17,    /* (nv,)    X_i^2 | Y                          (:368) */
    LCX_M_SYN_XIYJ = 18,   /* (nv, m)  X_i Y_j = X^T.Y / N                (:355) */
    LCX_M_CY = 19,         /* (m, m)   cov(y)                             (:356) */
    LCX_M_YJ2 = 20         /* (m,)     Y_j^2 = diag(cy)                   (:357) */
} lcx_moment_key;

/* ---- library ------------------------------------------------------------------------------ */
int         lcx_abi_version(void);
const char* lcx_last_error(void);
int         lcx_device_count(int* out_count);

/* ---- handle -------------------------------------------------------------------------------- */
/* Replaces cm.cublas_init() (:85-86).  Allocates all device state for an
 * (n_samples x nv_local) shard with n_hidden factors on HIP device `device`.  n_hidden <= 1024: padded to 16 / 32 / 64 / 128 /
 * 256 columns on the tuned kernels, to 512 / 1024 on the wide path (257+ factors: every contraction on one generic LDS-staged MFMA
 * GEMM with the factor axis tiled like any other, one thread per factor in the per-variable kernels - correct, untuned).
 * A failed allocation releases what was allocated before it.
 * How the shard is kept resident follows from its shape (lcx_x_layout): large shards - both X-streaming contractions on the stream-K
 * kernels - as ONE panel-major copy that serves both at full speed (a 50 000 x 1 000 000 float32 problem, BASELINE configs[3]
 * unsharded, fits one 288 GB MI355X); small shards row-major plus a transposed copy, so that each wave-split pass reads its big
 * operand along its preferred axis.  lcx_kernel_name(h, 0 | 1 | 2, ..) names the kernels in use; lcx_bytes_resident reports the bytes. */
int lcx_create(lcx_ctx** out, int64_t n_samples, int64_t nv_local, int n_hidden,
               int dtype, int device);
int lcx_destroy(lcx_ctx* h);
/* stream all work is enqueued on (a hipStream_t); NULL = the handle's own stream. */
int lcx_set_stream(lcx_ctx* h, void* hip_stream);
int lcx_synchronize(lcx_ctx* h);

/* Number of ranks that share the n_variables axis (default 1).  With 1 the *_a/_b/_c halves of a
 * level need no exchange in between, and the engine folds their tiny tail kernels together
 * (lcx_moments_c and lcx_update_d then launch nothing). */
int lcx_set_world(lcx_ctx* h, int world);

/* enable != 0 (default): lcx_update_c runs both passes of _sig (:210-211) and leaves sig_grad and the direction in
 * X^T.Y space, which the linear trial mode (lcx_trial_linear_*) needs.  0: only the first pass - update_tangent
 * (:305), the one thing the reference uses sig_grad for, is formed in Y space:
 *     (1-eps^2)/N <Y_g, X.update^T> + eps^2 <grad, update>,   X.update^T = -rj (Y_g - c Y)
 * (LCX_M_SIG_GRAD is then not produced).  Callers that only use lcx_make_trial save one pass over X per update. */
int lcx_set_linear_mode(lcx_ctx* h, int enable);
/* Force the exchange steps on with one rank (the all-reduces a caller issues are then identities): runs the
 * multi-rank code path of the library on a single GPU.  lcx_set_world(h, n > 1) enables it by itself. */
int lcx_set_exchange(lcx_ctx* h, int enable);

/* Exchange buffers (device pointers).  ybuf, elements of the working dtype:
 *     [ Y: n_samples_padded * m_padded | tail: m_padded^2 | Y_g: n_samples_padded * m_padded, only on shards that can run the
 *       merged pass (lcx_timing_read kind 2) ]
 * A caller that owns the exchange all-reduces the first two parts (n_pad * m_pad + m_pad^2 elements, lcx_geometry) between
 * the levels; the third part is only ever exchanged by the library itself.  sbuf: lcx_sbuf_count doubles (lcx_read_sbuf).
 * lcx_exchange_layout reports the element counts of the whole buffers; lcx_bind_exchange(NULL, NULL) restores the handle's
 * own buffers. */
int lcx_exchange_layout(lcx_ctx* h, int64_t* ybuf_elems, int64_t* sbuf_elems,
                        void** ybuf_dev, void** sbuf_dev);
int lcx_bind_exchange(lcx_ctx* h, void* ybuf_dev, void* sbuf_dev);

/* ---- exchange inside the library --------------------------------------------------------------------------------------
 * The sums over all variables (Y = X.W^T with W.W^T, the TC sums with H and the tangent, Y_g with Bj - the
 * reference's :247, :259, :294, :301-305 seen from one shard) are all-reduces of the two exchange buffers.  Once a
 * transport is bound, every level below issues the all-reduce of what it produced ITSELF, on the handle's stream, right
 * behind the kernels that wrote it - the caller calls the same levels and exchanges nothing - and lcx_iterate serves
 * several ranks: its line-search decisions are taken from all-reduced scalars, which are bit-identical on every rank,
 * so every rank issues the same sequence of collectives without talking about it.
 *
 *   RCCL (one communicator per handle, xGMI inside a node):
 *       rank 0:     lcx_comm_unique_id(id)         ncclGetUniqueId; ship the LCX_COMM_ID_BYTES to the other ranks
 *       every rank: lcx_comm_init(h, n, rank, id)  ncclCommInitRank (collective); implies lcx_set_world(h, n)
 *   any other transport: lcx_set_exchange_hook(h, fn, user): fn sums `count` elements of `dtype` (LCX_F32 / LCX_F64) at the
 *       device address `dev_buf` over the ranks, in place, ordered with `hip_stream`; returns 0 on success.
 * lcx_exchange_info: kind -1 = no exchange steps (one rank), 0 = the caller exchanges between the levels, 1 = RCCL, 2 = hook.
 *
 * The hook's `hip_stream` is the stream the collective must be ordered with: the handle's stream - or, with the environment's
 * LCX_Y_PIPELINE=chunks[:n] (default off, read by lcx_create), a second stream of the library's: the N x m all-reduces of
 * lcx_moments_a (Y, :247) and lcx_update_b (Y_g, :210) then go out in n (default 4, at most 16) row chunks, each behind the event of
 * its chunk's slot reduction, and on small shards (the wave-split kernels) behind its own row chunk of the PASS, so that the exchange
 * of chunk c overlaps the pass of chunk c+1 (:247 -> :259 is where a latency-exposed shard waits) - as long as a chunk's launch
 * still fills the chip ("chunks:n:pass" forces it on any shape).  Every element is summed over
 * slots and ranks as without chunks; every rank issues the same chunks in the same order; the m x m tail rides in the last chunk.
 * LCX_Y_PIPELINE=signal[:n[:poll]]: the same exchange with ONE launch of the wave-split pass - it stores its partial tiles
 * write-through and the launch's epoch into a signal word per row chunk as the chunk's last tile lands; the second stream waits on the
 * word (hipStreamWaitValue32; ":poll", or a runtime without wait-value: a one-lane polling kernel, bounded), sums the chunk's slots
 * with the unpipelined reduction kernel (same bits) and hands the chunk to the transport while the pass goes on: 5 us per extra
 * chunk instead of 22, the pass itself within 1.5 % of the plain one. */
#define LCX_COMM_ID_BYTES 128
typedef int (*lcx_allreduce_fn)(void* user, void* dev_buf, int64_t count, int dtype, void* hip_stream);
/* local, no collective: LCX_OK iff this process can bind librccl (dlopen).  ncclCommInitRank is collective - the ranks compare
 * this answer BEFORE any of them enters lcx_comm_init, so that a rank without the library cannot leave the others blocked in it */
int lcx_comm_probe(void);
int lcx_comm_unique_id(void* id_out /* LCX_COMM_ID_BYTES */);
int lcx_comm_init(lcx_ctx* h, int nranks, int rank, const void* id /* LCX_COMM_ID_BYTES */);
int lcx_set_exchange_hook(lcx_ctx* h, lcx_allreduce_fn fn, void* user);
int lcx_exchange_info(lcx_ctx* h, int* kind, int* world, int64_t* allreduces_issued);
/* First contact with a bound transport (collective: every rank calls it with its own rank in [0, world), right after lcx_comm_init / lcx_set_exchange_hook and
 * lcx_bind_exchange, before any level): all-reduces the Y exchange buffer at its real size on the handle's stream, once with a
 * rank-dependent integer pattern whose sum is known in closed form and once with rank-dependent values over 12 binades (with
 * LCX_Y_PIPELINE=chunks that second one on the library's second stream, behind an event, as the pipelined exchange issues its
 * chunks - the stream argument of a hook is part of what is tested), then shares the verdicts through the scalar buffer.  Whatever can fail on one rank alone (arguments, an allocation) happens first and is
 * shared by a one-element all-reduce that every rank enters: a local failure returns an error on EVERY rank instead of leaving the
 * others blocked inside the big all-reduce.  LCX_OK and *ok = 1 iff every rank got the right sums AND all ranks hold the same
 * bits (lcx_iterate's decisions rely on that); LCX_ERR_COMM with the diagnosis otherwise, on every rank alike.
 * seconds_per_allreduce (may be NULL): host wall time of one Y-buffer all-reduce.  The sums the reference
 * forms in one address space (:247, :259) are only as good as this exchange. */
int lcx_comm_selftest(lcx_ctx* h, int rank, int* ok, double* seconds_per_allreduce);

/* ---- data ---------------------------------------------------------------------------------- */
/* Replaces cm.CUDAMatrix(x) (:427-428): upload the preprocessed shard, row-major, leading
 * dimension ld (elements of the working dtype). */
int lcx_upload_x(lcx_ctx* h, const void* x_host, int64_t ld);
/* Replaces preprocess(x, fit) + cm.CUDAMatrix(x) (:397-429) for a RAW shard: upload, then on the device
 * impute missing cells by the column mean of the observed cells (mean_impute, :497-510; a cell is missing if
 * it is NaN or equals `missing`, only when has_missing != 0), estimate theta = (mean, std) (fit != 0;
 * 'standard': std over the observed cells :411-413, 'outliers': np.std over all rows :420-421, both clipped
 * at 1e-10) or take the given one (fit == 0), standardise, and for kind 2 squash the tails with g (:483-487).
 * kind: 0 pass-through ('none' and any unknown name, :404-405), 1 'standard', 2 'outliers', 3 'empirical' (:424-426: every
 * column becomes norm.ppf((rankdata(column) - 0.5) / n_samples), ties get their average rank; no theta - a per-column
 * segmented sort of a transposed copy of the columns, so a new batch is transformed by uploading it into a handle of its own).
 * Every step is per column: a panel-major shard is filled through row-major staging blocks of columns, each preprocessed on its own.
 * mean_io / std_io: nv_local values of the working dtype; n_obs_out: nv_local int64 or NULL;
 * max_abs_out: max |x~| (for the "more than 6 stds" warning, :416-417) or NULL. */
int lcx_upload_preprocess(lcx_ctx* h, const void* x_raw_host, int64_t ld, int kind, int has_missing, double missing,
                          int fit, void* mean_io, void* std_io, int64_t* n_obs_out, double* max_abs_out);
/* On-device synthetic shard for sizes that cannot be staged on the host (SURVEY.md 8d Gen-A/B):
 * element (row, col_offset+col) of a counter-based N(0,1) generator keyed by seed; kind 0 = iid,
 * kind 1 = planted groups (n_groups latent factors + unit noise).  Columns are standardised on
 * device (mean 0, variance 1 over samples, as preprocess 'standard' does, :409-415). */
int lcx_generate_x(lcx_ctx* h, uint64_t seed, int kind, int n_groups, int64_t col_offset);
/* copy the resident shard back (tests) */
int lcx_download_x(lcx_ctx* h, void* x_host, int64_t ld);

/* self.ws (m x nv_local, row-major) */
int lcx_set_ws(lcx_ctx* h, const void* w_host);
int lcx_get_ws(lcx_ctx* h, int which, void* w_host);
/* ws[order] (:162) */
int lcx_permute_factors(lcx_ctx* h, const int32_t* order);

/* ---- moments: _calculate_moments_ns (:236-275), split at the sums over all variables ------- */
/* a: Y_partial = X_shard . W_shard^T (:247) and W_shard.W_shard^T  -> ybuf                      */
int lcx_moments_a(lcx_ctx* h, int which);
/* b: (ybuf now holds the global sums) uj (:248-249), early-exit flag (:250-251), X^T.Y (:259),
 *    rho (:260), ry (:261,:263), invrho, rhoinvrho, Qij, Si, Qi-Si^2 (:264-269), the two per-shard
 *    log sums of TC (:272-273) -> sbuf[0..1], a pending update_tangent -> sbuf[2], and the H partial of this set
 *    (:294) -> sbuf[8..).  With one GPU the same call forms TC and publishes the state (lcx_moments_c is a no-op)  */
int lcx_moments_b(lcx_ctx* h, int which, double eps, int quick);
/* c: (sbuf[0..2] global) TC (:272-274) and update_tangent (:305) -> state scalars of the set     */
int lcx_moments_c(lcx_ctx* h, int which);

/* detail part (:277-287): per-shard sums -> the detail range of sbuf (see lcx_read_sbuf), m+3 values:
 *   [0..m) sum_i MI_ji, [m] sum_i max_j MI_ji, [m+1] sum_i I(X_i;Y), [m+2] sum_ij MI_ji          */
int lcx_moments_detail(lcx_ctx* h, int which);

/* ---- update: _update_ns (:290-305) ---------------------------------------------------------- */
/* a: H partial (:294) of set 0 -> sbuf[8 .. 8+m_padded^2).  lcx_moments_b / lcx_trial_linear_b already leave the H
 *    of the set they evaluate there, so a fit only needs this call to restore it after a discarded trial.   */
int lcx_update_a(lcx_ctx* h);
/* b: grad (:296-300), Bj partial (:302), Y_g partial = X.grad^T (first half of _sig, :210)
 *    -> ybuf (Bj in the tail).  Merged form (see lcx_timing_read, kind 2): Bj, update (:303) and ws + update (:320) are
 *    formed first, then ONE pass over X yields Y_g (kept in a buffer of its own) and the Y of the eta = 1 trial (-> ybuf and
 *    set 1); lcx_update_c then only adds the Y-space half, and lcx_make_trial(1.0) + lcx_moments_a(1) launch nothing.    */
int lcx_update_b(lcx_ctx* h, double eps);
/* c: X^T.Y_g, sig_grad (:211-212), update (:303), per-block partials of update_tangent (:305).  They are summed
 *    into sbuf[2] by the tail of the NEXT lcx_moments_b / lcx_trial_linear_b (the first trial of the direction), which
 *    with one GPU also stores the tangent in the state scalars of set 0 and of the trial; lcx_read_state(0) before any
 *    trial sums them on demand                                                                          */
int lcx_update_c(lcx_ctx* h, double eps);
/* d: marks the direction as ready (several ranks: the tangent becomes global with the first trial's scalar
 *    all-reduce and lcx_moments_c stores it in the trial's state scalars)                             */
int lcx_update_d(lcx_ctx* h);
/* w_update = ws + eta*update (:320) into set 1                                                   */
int lcx_make_trial(lcx_ctx* h, double eta);
/* Linear trial mode (optional).  X^T.(X.u^T) is linear in u and acts per factor, and
 * update_j is a per-factor combination of grad_j and ws_j, so Y and X^T.Y of ws + eta*update follow
 * from quantities lcx_update_c already has - a trial then costs no pass over X.  Same outputs as
 * lcx_make_trial + lcx_moments_a/_b (set 1) up to rounding.
 * a: w_update (:320), W.W^T partial -> tail of ybuf, Y' = Y + eta*Y(update)
 * b: (tail of ybuf global) uj, early-exit flag, rho ... Qi-Si^2, TC partial sums -> sbuf[0..1]   */
int lcx_trial_linear_a(lcx_ctx* h, double eta);
int lcx_trial_linear_b(lcx_ctx* h, double eps, double eta);
/* self.ws, self.moments = w_update, m_update (:139,:334): swap sets                              */
int lcx_accept_trial(lcx_ctx* h);

/* ---- one whole iteration (:290-334) in one call -------------------------------------------------------------
 * `_update_ns` with its back-tracking line search, the host side of the decisions included: the levels above
 * (lcx_update_b/_c, lcx_make_trial, lcx_moments_a/_b per trial, lcx_accept_trial) are sequenced inside the library, the
 * scalars of each trial are read from the pinned mirror and the next launches follow immediately - no interpreter between
 * a trial's result and the work that depends on it.  tc_cur = TC of the current solution (LCX_S_TC of set 0).
 * more != 0: the caller intends to iterate again unless |TC_new - tc_cur| < tol (:152); the direction and the first trial
 * of the NEXT iteration are then enqueued before this call returns, so that the GPU works while the caller does its
 * book-keeping (:151, :166-175).  Any other state-changing entry point abandons that work safely.
 * out8: [0] status 0 = accepted (:334), 1 = update_tangent >= 0, nothing changed (:306-311), 2 = the step size underflowed
 *            on an invalid trial (the reference then returns m_update = False, :144-149);
 *       [1] TC of the accepted trial, [2] update_tangent (:305), [3] trials evaluated (:321), [4] trials with max uj >= 1
 *       (:322-326), [5] 1 if the step size fell below min(tol, 1e-10) (:316-319), [6] moment evaluations waited for,
 *       [7] 1 if the next iteration was started.
 * Several ranks: needs the exchange inside the library (lcx_comm_init / lcx_set_exchange_hook); every rank calls it with the same
 * arguments and gets the same answer.  While the caller owns the exchange (neither bound) it sequences the levels itself and
 * this call returns LCX_ERR_STATE. */
int lcx_iterate(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out8);
/* enable != 0: inside lcx_iterate the back-tracking trials AFTER the first one of an iteration (:320-321 at eta = 1/2, 1/4, ...)
 * take X.w_update^T by linearity - Y + eta X.update^T, both already computed exactly by this iteration (:247 of the accepted
 * solution, :210 through lcx_update_c) - and make ONE pass over X (X^T.Y', :259) instead of two.  Same mathematics as :321, other
 * rounding (an exact re-association like the Y-space update_tangent of lcx_set_linear_mode(0)); nothing is carried across
 * iterations but the Y of an accepted solution, which each step mixes convexly with fresh products (no drift, no re-anchoring).
 * Off by default: the default iteration re-evaluates every trial with two passes, as the reference does. */
int lcx_set_trial_reuse(lcx_ctx* h, int enable);
/* The reference divides every sample moment by `self.n_samples`, the sample count of the FIT (linearcorex.py:249, :260, :228, :211,
 * :355) - also when `transform(x_new, details=True)` (:392-394) hands `_calculate_moments` a batch of another row count.  A handle
 * that holds such a batch (x_new preprocessed with the fitted theta, the fitted W) is told the fit's count here; every level then
 * divides by it instead of the handle's own n_samples (the default). */
int lcx_set_sample_divisor(lcx_ctx* h, double n_samples);

/* ---- synergistic branch: discourage_overlap=False (:336-384) -------------------------------------
 * One evaluation of _calculate_moments_syn on set `which`:
 *   lcx_moments_a(which)              Y_partial = X_shard . W_shard^T (:347)            -> ybuf     | all-reduce ybuf
 *   lcx_syn_moments_b(which, yscale)  cy, Y_j^2, ry (:356-358; cy = Y^T.Y/N + yscale^2 I, which equals
 *                                     ws.dot(X_i Y_j) + yscale^2 I because W.X^T = Y^T - no extra collective),
 *                                     X^T.Y (:355), rho (:359), X_i Z_j (:367), X_i^2|Y (:368), per-shard sums
 *                                     [0..m) sum_i MI_ji, [m] -, [m+1] sum_i I(X_i;Y), [m+2] sum_ij MI -> detail range of sbuf | all-reduce it
 *   lcx_syn_moments_c(which)          TC = sum_i I(X_i;Y) - sum_j I(Y_j;X) (:373) -> state scalars
 * One _update_syn (:375-383) from set 0 into the weights of set 1 (then the moments above with which=1 and
 * lcx_accept_trial):
 *   lcx_syn_update_a()                H partial (:378) -> sbuf[8..8+m_padded^2)                     | all-reduce
 *   lcx_syn_update_b(eta)             ws' = (1-eta) ws + eta (X_i Z_j^T / X_i^2|Y - H ws) (:380-382)          */
int lcx_syn_moments_b(lcx_ctx* h, int which, double yscale);
int lcx_syn_moments_c(lcx_ctx* h, int which);
int lcx_syn_update_a(lcx_ctx* h);
int lcx_syn_update_b(lcx_ctx* h, double eta);
/* get_covariance, synergistic branch (:452-455): rows [row0, row0+nrows) of X_i Z_j . X_i Y_j^T, diagonal 1,
 * scaled by std_i std_k */
int lcx_covariance_rows_syn(lcx_ctx* h, const void* std_host, int64_t row0, int64_t nrows, void* out_host);

/* ---- stage change (:129-133) ---------------------------------------------------------------- */
int lcx_rescale_ws(lcx_ctx* h, double eps_old, double eps_new);
/* ws /= (10 * _norm(x, ws)) (:117); needs lcx_moments_a + exchange done for set 0 with eps=0    */
int lcx_init_scale_ws(lcx_ctx* h);

/* ---- readback -------------------------------------------------------------------------------- */
/* the LCX_S_COUNT scalars of a set; waits (polling a pinned host mirror the last kernel of a level
 * writes) until the work enqueued so far for that set has published them */
int lcx_read_state(lcx_ctx* h, int which, double* out);
/* host_out: the reference-shaped array, working dtype (see lcx_moment_key) */
int lcx_get_moment(lcx_ctx* h, int which, int key, double eps, void* host_out);
/* upload a moment of set `which` (only LCX_M_RHOINVRHO and LCX_M_SI: what get_covariance needs,
 * :447) - used to restore a pickled model (vis_corex.py:549-551) without refitting */
int lcx_set_moment(lcx_ctx* h, int which, int key, const void* host_in);
/* synchronise and copy `count` doubles of the scalar exchange buffer starting at `offset`.  Layout (doubles):
 *   [0] sum_i log(1+Si), [1] sum_i log(1+Qi-Si^2), [2] update_tangent partial, [3..8) spare,
 *   [8, 8+m_padded^2)                       H partial (:294 / :378) of the moment set evaluated last,
 *   [8+m_padded^2, 8+m_padded^2+m+3)        detail sums (lcx_moments_detail, lcx_syn_moments_b).
 * A multi-GPU caller all-reduces [0, 8+m_padded^2) once per moment evaluation (between _b and _c) - that carries the
 * TC sums, the tangent of the direction and the H the next update needs - and the detail range after the calls
 * that produce it. */
int lcx_read_sbuf(lcx_ctx* h, int64_t offset, int64_t count, double* out);

/* ---- outputs ---------------------------------------------------------------------------------- */
/* get_covariance (:443-451), rows [row0, row0+nrows) of the nv_local x nv_local matrix;
 * std_host = theta[1] (nv_local, working dtype); out_host is nrows x nv_local row-major. */
int lcx_covariance_rows(lcx_ctx* h, double eps, const void* std_host, int64_t row0, int64_t nrows,
                        void* out_host);
/* get_covariance (:443-455), the whole nv_local x nv_local matrix into out_host (row-major, leading dimension
 * ld_out >= nv_local elements): synergistic == 0 the branch of :446-451 (needs the moments of set 0), != 0 the branch
 * of :452-455 (needs lcx_syn_moments_b).  Row blocks are produced on the device (rank-m_padded product on MFMA),
 * copied through two pinned staging blocks and placed into out_host while the next block is being computed.
 * kernel_seconds (may be NULL): device time of the product kernels alone, from HIP events. */
int lcx_covariance(lcx_ctx* h, int synergistic, double eps, const void* std_host, void* out_host, int64_t ld_out,
                   double* kernel_seconds);
/* device bytes owned by the handle; x_bytes: the part that is the resident shard (see lcx_x_layout) */
int lcx_bytes_resident(lcx_ctx* h, int64_t* total, int64_t* x_bytes);
/* How the preprocessed shard x~ (what cm.CUDAMatrix(x) holds in the reference, :427-428) is resident - chosen by lcx_create from the
 * shard's shape, invisible in every result:
 *   2  ONE panel-major copy [n_variables / P][n_samples][P], P = 64 bytes of a row: large shards, both X passes (:247, :259) on the
 *      stream-K kernels at full speed from the same bytes (LCX_X_LAYOUT=panel forces it, =rows forbids it);
 *   0  row-major + a transposed copy: small shards (each pass reads the copy whose contiguous axis it does not contract);
 *   1  row-major only: models with more than 256 factors, or LCX_SINGLE_COPY=1. */
int lcx_x_layout(lcx_ctx* h, int* layout);
/* Arithmetic of the two X-streaming contractions (:247 / :210 x.dot(ws.T), :259 / :211 x.T.dot(y)) of a FLOAT32 shard in the
 * panel-major layout; switchable at any time between two launches, results of both modes within float32 rounding of each other:
 *   0  (default) v_mfma_f32_16x16x4_f32: exact float32 products, float32 accumulation - the float32 MATRIX rate of gfx950 equals
 *      its float32 VECTOR rate (157 TF/s), 1/16 of the bf16 rate, and bounds these passes;
 *   1  every operand element split exactly into three bf16 numbers by rounding to nearest (hi = bf16(x), mid = bf16(x - hi),
 *      lo = x - hi - mid: 8 + 8 + 8 significand bits, signed residuals), 6 of the 9 partial products (all terms down to 2^-16 of the
 *      product; the dropped ones are zero-mean, 2^-27 of it in the rms and at most 2^-24.4 - below half a float32 rounding of the
 *      product) accumulated in float32 by v_mfma_f32_16x16x32_bf16 - 2.5 x less matrix-pipe time, the passes become HBM / power
 *      bound (1.35-1.45 x the fit iterations per second at the config-3 / config-4 shards).  Error against a float64 contraction:
 *      1.0-1.2 x that of mode 0 (profiles/r04_gemm_probe9_rne.txt); every parity fixture holds at the float32 bars in both modes.
 *      (Round 4 split by truncation - 6 % faster, but every dropped term then has the sign of its product, a systematic shrink of
 *      up to 2^-21.3; the library ships the unbiased split.  Operands within half a bf16 ulp of FLT_MAX would round to infinity.)
 *      An opt-in: never the arithmetic behind a reported `value` (bench.py), which stays mode 0.
 * Mode 1 needs layout 2 and 32 / 64 / 128 padded factors; on any other handle the call succeeds and leaves mode 0 (read it back with
 * lcx_f32_gemm).  Environment default: LCX_F32_GEMM=split.  Costs 6 bytes per element of the small operand (one scratch buffer). */
int lcx_set_f32_gemm(lcx_ctx* h, int mode);
int lcx_f32_gemm(lcx_ctx* h, int* mode);
/* transform (:386-395): out (n_rows x m) = x (n_rows x nv_local, ld) . ws^T  (per-shard partial) */
int lcx_project(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, void* out_host);

/* transform (:386-395) of RAW rows without missing values: standardise with theta on the device (kind as
 * above), then x~ . ws^T */
int lcx_project_raw(lcx_ctx* h, const void* x_raw_host, int64_t n_rows, int64_t ld, int kind, const void* mean,
                    const void* std, void* out_host);

/* predict (:440-441): out (n_rows x nv_local, row-major, leading dimension ld_out) = invert(y . X_i Z_j^T) for
 * y_host (n_rows x n_hidden, row-major): the rank-m_padded product on MFMA in row blocks with `invert` (:431-438) as its
 * epilogue - out = std_c f(.) + mean_c, f = identity (kind 1 'standard') or g_inv (:490-494; kind 2 'outliers'); kind 0
 * returns the product itself - staged to the host like lcx_covariance.  X_i Z_j: xz_host (nv_local x n_hidden, row-major) if
 * given (a model restored from a pickle), else the resident moments of set 0 (synergistic == 0: solve(ry, rho)^T of :280;
 * != 0: the X_i Z_j of lcx_syn_moments_b, :367).  kernel_seconds (may be NULL): device time of the product kernels. */
int lcx_predict(lcx_ctx* h, const void* y_host, int64_t n_rows, int synergistic, const void* xz_host, int kind, const void* mean,
                const void* std, void* out_host, int64_t ld_out, double* kernel_seconds);
/* invert (:431-438) of host rows x (n_rows x nv_local, ld): out = std_c f(x) + mean_c (kind / theta as above) */
int lcx_invert(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean, const void* std,
               void* out_host, int64_t ld_out);

/* ---- measurement ------------------------------------------------------------------------------ */
/* HIP-event timing sites (kinds) on the stream that carries the work:
 *   the X-streaming GEMM kernels - kind 0 = X.B^T ("nt", :247/:210), kind 1 = X^T.Y ("tn", :259/:211), kind 2 = the merged pass
 *   X.[grad | ws + update]^T (the :210 pass and the first trial's :321 pass as one launch with 2 x m_padded columns; float32
 *   shards on the column-tiled kernel with <= 64 padded factors - see lcx_update_b);
 *   the exchange steps of a handle with a bound transport (the sums of :247, :259, :294, :301-305 over the ranks) - kind 3 = the
 *   Y-buffer all-reduce of lcx_moments_a ([Y | W.W^T], every chunk of the pipelined form), kind 4 = the one of lcx_update_b
 *   ([Y_g | Bj]; merged form [Y' | W'.W'^T | Y_g]), kind 5 = the scalar buffer of an evaluation (TC sums, tangent, H), kind 6 = the
 *   small ones (Bj in front of the merged pass, W'.W'^T of a trial taken by linearity, a restored H).  With RCCL the pair
 *   brackets the all-reduce kernel - including its wait for a slower rank; with a host-blocking hook the gap it leaves on the
 *   stream.  Not timed: first contact (lcx_comm_selftest reports its own figure) and the one-flag agreement on the merged pass. */
int lcx_timing_enable(lcx_ctx* h, int enable);
/* time only every `every`-th launch of each site (an event pair costs ~5 us of stream time; default 1 = all) */
int lcx_timing_sample(lcx_ctx* h, int every);
/* launches / total_ms of site `kind` in [0, 7): the timed launches that did their work.  An X pass whose trial went invalid
 * (:250-251) returns at its skip flag; such launches (shorter than a fifth of the longest of their kind) are not averaged in */
int lcx_timing_read(lcx_ctx* h, int kind, int64_t* launches, double* total_ms);
/* every launch of site `kind` issued since the last reset while timing was enabled (timed or skipped by the sampling) */
int lcx_timing_passes(lcx_ctx* h, int kind, int64_t* passes);
int lcx_timing_reset(lcx_ctx* h);
/* geometry actually used (for the roofline arithmetic): padded sizes and launch shapes */
int lcx_geometry(lcx_ctx* h, int64_t* n_pad, int64_t* ldx, int* m_pad, int64_t* info8);

/* name of the kernel function behind pass `kind` as rocprofv3 prints it (without "void " and the
 * argument list), so that bench.py's roofline line and the committed rocprof summary name the same row */
int lcx_kernel_name(lcx_ctx* h, int kind, char* buf, int64_t len);

#ifdef __cplusplus
}
#endif
#endif /* LCX_H */
