// lcx_core.hip - handles, streams, the exchange transport and its first-contact self-test, state readback, timing (include/lcx.h).
#include "engine.hpp"

extern "C" {

int lcx_abi_version(void) { return 1; }

const char* lcx_last_error(void) { return g_err.c_str(); }

int lcx_device_count(int* out_count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (out_count) *out_count = n;
    return LCX_OK;
}

int lcx_create(lcx_ctx** out, int64_t n_samples, int64_t nv_local, int n_hidden, int dtype, int device) {
    if (!out || n_samples < 1 || nv_local < 1 || n_hidden < 1) return fail(LCX_ERR_ARG, "lcx_create: bad sizes");
    if (dtype != LCX_F32 && dtype != LCX_F64) return fail(LCX_ERR_ARG, "lcx_create: dtype must be LCX_F32 or LCX_F64");
    const int ct = ct_for(n_hidden);
    if (!ct) return fail(LCX_ERR_ARG, "lcx_create: n_hidden > 1024 is not supported by this build");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LCX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(LCX_ERR_ARG, "lcx_create: device index out of range");
    HIPCHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device));

    lcx_ctx* h = new lcx_ctx();
    h->split = false;
    h->bsp = nullptr;
    h->bsp_bytes = 0;
    h->device = device;
    h->dtype = dtype;
    h->es = dtype == LCX_F32 ? 4 : 8;
    h->N = n_samples;
    h->Ndiv = (double)n_samples;
    h->V = nv_local;
    h->M = n_hidden;
    h->CT = ct;
    h->Mp = 16 * ct;
    h->Npad = round_up(n_samples, 64);
    h->ldx = round_up(nv_local, 64);        // 64 elements: whole 128 B chunks and whole tn column tiles
    h->timing = false;
    h->t_every = 1;
    for (int k = 0; k < LCX_T_KINDS; ++k) { h->t_count[k] = 0; h->t_launch[k] = 0; h->t_pass[k] = 0; h->t_ms[k] = 0.0; h->t_max[k] = 0.0; h->t_skipped[k] = 0; }
    h->have_direction = false;
    h->target_waves = prop.multiProcessorCount * 12;
    h->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(LCX_ERR_HIP, "lcx_create: cannot create a stream");
    }
    h->stream = h->own_stream;
    hipStream_t st = h->stream;
    {
        // One resident copy of the shard instead of two: X.B^T then reads X itself (gemm_cr: 4-6 % slower than gemm_ct on the
        // transposed copy, tools/gemm_probe4 cr).  LCX_SINGLE_COPY=1 / 0 forces; by default only when two copies would not
        // leave room for the rest (moments and work space are ~ 20 M x V arrays).
        const char* e = getenv("LCX_SINGLE_COPY");
        const double xb = (double)h->Npad * (double)h->ldx * (double)h->es;
        size_t free_b = 0, total_b = 0;
        bool want = false;
        if (e && *e) want = atoi(e) != 0;
        else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            want = 2.0 * xb + 24.0 * (double)h->ldx * h->Mp * h->es > 0.94 * (double)free_b && xb < 0.9 * (double)free_b;
        h->single_copy = want || ct > 16;     // the wide path reads X.B^T from the row-major copy anyway (gemm_wide)
        (void)hipGetLastError();
    }

    int rc = LCX_OK;
    {
        rc = lcx_engine_geometry(h);
        if (rc != LCX_OK) { (void)lcx_destroy(h); return rc; }
    }
    const size_t es = h->es;
    const size_t mv = (size_t)h->ldx * h->Mp * es;
    const size_t vv = (size_t)h->ldx * es;
    const int Mp = h->Mp;
    // a failed allocation releases everything allocated so far (lcx_destroy copes with a half-built handle)
#define A_(ptr, bytes) do { const size_t b_ = (bytes); int r_ = dev_alloc((void**)&(ptr), b_, st); if (r_ != LCX_OK) { (void)lcx_destroy(h); return r_; } h->bytes_resident += b_ ? b_ : 16; } while (0)
    A_(h->X, (size_t)h->Npad * h->ldx * es);
    if (!h->single_copy && !h->panel) A_(h->XT, (size_t)h->Npad * h->ldx * es);
    for (int k = 0; k < 2; ++k) {
        A_(h->Wt[k], mv);
        A_(h->set[k].Y, (size_t)h->Npad * Mp * es);
        A_(h->set[k].D, mv);
        A_(h->set[k].rho, mv);
        A_(h->set[k].rir, mv);
        A_(h->set[k].qij, mv);
        A_(h->set[k].si, vv);
        A_(h->set[k].q2, vv);
        A_(h->set[k].hscale, vv);
        A_(h->set[k].uj, sizeof(double) * Mp);
        A_(h->set[k].ry, sizeof(double) * Mp * Mp);
        A_(h->set[k].wmag, sizeof(double) * Mp);
        h->set[k].xz = h->set[k].x2y = nullptr;
        h->set[k].cy = h->set[k].yj2 = h->set[k].inv_sd = nullptr;
    }
    A_(h->grad, mv);
    A_(h->update, mv);
    A_(h->sgrad, mv);
    A_(h->scratch, mv);
    A_(h->ydir, (size_t)h->Npad * Mp * es);
    A_(h->ddir, mv);
    h->gw = h->y2part = h->ygbuf = nullptr;
    h->y1_ready = false;
    h->bjg = nullptr;
    if (h->merged_ok) {
        A_(h->gw, 2 * mv);
        A_(h->y2part, (size_t)h->nt2_S * h->Npad * 2 * Mp * es);
        A_(h->bjg, (size_t)Mp * es);
    }
    h->have_linear = false;
    h->full_sig = true;
    h->exchange = false;
    h->w1_ready = h->y1_ready = false;
    h->ybuf_main = h->Npad * Mp + (int64_t)Mp * Mp;
    h->ybuf_elems = h->ybuf_main + (h->merged_ok ? h->Npad * Mp : 0);    // Y_g of the merged pass right behind the tail
    h->sbuf_elems = (int64_t)SB_H + (int64_t)Mp * Mp + Mp + 8;
    A_(h->ybuf_own, (size_t)h->ybuf_elems * es);
    A_(h->sbuf_own, sizeof(double) * h->sbuf_elems);
    h->ybuf = h->ybuf_own;
    h->sbuf = h->sbuf_own;
    if (h->merged_ok) h->ygbuf = (char*)h->ybuf + (size_t)h->ybuf_main * es;
    A_(h->ypart, h->nt_S > 1 ? (size_t)h->nt_S * h->Npad * Mp * es : 16);
    A_(h->dpart, (size_t)h->tn_S * mv);
    {
        const int gs = h->gn_S > h->gv_S ? h->gn_S : h->gv_S;
        A_(h->gpart, (size_t)gs * Mp * Mp * es);
        A_(h->gpartw, (size_t)gs * Mp * Mp * es);
    }
    A_(h->tcpart, sizeof(double) * 2 * 2048);
    A_(h->bjpart, sizeof(double) * Mp * 2048);
    A_(h->tanpart, sizeof(double) * 2048);
    A_(h->detpart, sizeof(double) * (Mp + 3) * 2048);
    A_(h->ryinv, sizeof(double) * Mp * Mp);
    A_(h->invwork, sizeof(double) * Mp * 2 * Mp);
    A_(h->states, sizeof(SetState) * 2);
    A_(h->order_dev, sizeof(int) * Mp);
    A_(h->ticket, 256);        // [0] small_moments; [16..25) moments_epilogue; [32..41) update_kernel
    h->merged_agreed = -1;
    h->ypipe = 0;
    {
        // "chunks" (4 row chunks), "chunks:n" (n <= 16), "chunks:n:pass" (per-chunk launches of the pass: tests);
        // "signal[:n[:poll]]": one pass launch with in-launch chunk signalling (engine.hpp, ypipe_signal)
        const char* e = getenv("LCX_Y_PIPELINE");
        h->ypipe_force_pass = h->ypipe_signal = h->ypipe_poll = false;
        const bool chunks = e && !strncmp(e, "chunks", 6), signal = e && !strncmp(e, "signal", 6);
        if (chunks || signal) {
            h->ypipe = e[6] == ':' ? atoi(e + 7) : 4;
            if (h->ypipe < 2) h->ypipe = 0;
            if (h->ypipe > 16) h->ypipe = 16;
            const char* f = e[6] == ':' ? strchr(e + 7, ':') : nullptr;
            h->ypipe_force_pass = chunks && f && !strcmp(f, ":pass");
            h->ypipe_signal = signal && h->ypipe > 1;
            h->ypipe_poll = signal && f && !strcmp(f, ":poll");
        }
    }
#undef A_
    h->set[0].st = h->states;
    h->set[1].st = h->states + 1;
    if (hipHostMalloc((void**)&h->host_states, sizeof(SetState) * 2, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)lcx_destroy(h);
        return fail(LCX_ERR_HIP, "lcx_create: cannot allocate the pinned state mirror");
    }
    memset(h->host_states, 0, sizeof(SetState) * 2);
    {
        SetState* dv = nullptr;
        if (hipHostGetDevicePointer((void**)&dv, h->host_states, 0) != hipSuccess) {
            (void)lcx_destroy(h);
            return fail(LCX_ERR_HIP, "lcx_create: no device address for the pinned state mirror");
        }
        for (int k = 0; k < 2; ++k) {
            h->set[k].hst = h->host_states + k;
            h->set[k].hst_dev = dv + k;
            h->set[k].seq_expect = 0;
        }
    }
    h->world = 1;
    h->n_exchanges = 0;
    h->seq_next = 0;
    h->spec_pending = h->spec_dirty = false;
    h->early_grad = h->grad_ready = false;
    h->spec_eps = 0.0;
    HIPCHECK(hipStreamSynchronize(st));
    {
        const char* e = getenv("LCX_F32_GEMM");           // "split": the bf16-pipe contractions where the shard supports them
        if (e && !strcmp(e, "split")) {
            const int rc2 = lcx_set_f32_gemm(h, 1);
            if (rc2 != LCX_OK) { (void)lcx_destroy(h); return rc2; }
        }
    }
    *out = h;
    return LCX_OK;
}

int lcx_destroy(lcx_ctx* h) {
    if (!h) return LCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm_stream) {
        (void)hipStreamSynchronize(h->comm_stream);
        for (auto& e : h->ypipe_ev) if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(h->comm_stream);
    }
    if (h->tr.kind == 1 && h->tr.comm) (void)rccl().CommDestroy(h->tr.comm);
    if (h->sig_counters) {
        (void)hipFree(h->sig_counters);
        for (auto& f : h->sig_flag) if (f) (void)hipFree(f);
    }
    void* ptrs[] = {h->X, h->XT, h->Wt[0], h->Wt[1], h->grad, h->update, h->sgrad, h->scratch, h->ydir, h->ddir, h->ybuf_own, h->sbuf_own,
                    h->gw, h->y2part, h->bjg, h->bsp,
                    h->ypart, h->dpart, h->gpart, h->gpartw, h->tcpart, h->bjpart, h->tanpart, h->detpart, h->ryinv, h->invwork,
                    h->states, h->order_dev, h->ticket};
    for (void* p : ptrs) (void)hipFree(p);
    for (int k = 0; k < 2; ++k) {
        MomentSet& s = h->set[k];
        void* q[] = {s.Y, s.D, s.rho, s.rir, s.qij, s.si, s.q2, s.hscale, s.uj, s.ry, s.wmag, s.xz, s.x2y, s.cy, s.yj2, s.inv_sd};
        for (void* p : q) (void)hipFree(p);
    }
    if (h->host_states) (void)hipHostFree(h->host_states);
    if (h->cov) {
        CovStage& c = *h->cov;
        for (int k = 0; k < 2; ++k) {
            if (c.dev[k]) (void)hipFree(c.dev[k]);
            if (c.pin[k]) (void)hipHostFree(c.pin[k]);
            if (c.ev_k[k]) (void)hipEventDestroy(c.ev_k[k]);
            if (c.ev_c[k]) (void)hipEventDestroy(c.ev_c[k]);
            if (c.t_a[k]) (void)hipEventDestroy(c.t_a[k]);
            if (c.t_b[k]) (void)hipEventDestroy(c.t_b[k]);
        }
        if (c.op_a) (void)hipFree(c.op_a);
        if (c.op_b) (void)hipFree(c.op_b);
        if (c.std_dev) (void)hipFree(c.std_dev);
        if (c.mean_dev) (void)hipFree(c.mean_dev);
        if (c.copy_stream) (void)hipStreamDestroy(c.copy_stream);
        delete h->cov;
    }
    for (auto& tp : h->pool) { (void)hipEventDestroy(tp.a); (void)hipEventDestroy(tp.b); }
    for (auto& tp : h->pending) { (void)hipEventDestroy(tp.a); (void)hipEventDestroy(tp.b); }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return LCX_OK;
}

int lcx_set_stream(lcx_ctx* h, void* s) {
    NEED_MUT(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return LCX_OK;
}

int lcx_synchronize(lcx_ctx* h) {
    NEED(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

int lcx_exchange_layout(lcx_ctx* h, int64_t* ye, int64_t* se, void** yd, void** sd) {
    NEED(h);
    if (ye) *ye = h->ybuf_elems;
    if (se) *se = h->sbuf_elems;
    if (yd) *yd = h->ybuf;
    if (sd) *sd = h->sbuf;
    return LCX_OK;
}

int lcx_bind_exchange(lcx_ctx* h, void* y, void* s) {
    NEED_MUT(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->ybuf = y ? y : h->ybuf_own;
    if (h->merged_ok) h->ygbuf = (char*)h->ybuf + (size_t)h->ybuf_main * h->es;
    h->sbuf = s ? (double*)s : h->sbuf_own;
    if (y) HIPCHECK(hipMemsetAsync(y, 0, (size_t)h->ybuf_elems * h->es, h->stream));
    if (s) HIPCHECK(hipMemsetAsync(s, 0, sizeof(double) * h->sbuf_elems, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

// ---- exchange inside the library ---------------------------------------------------------------------
int lcx_comm_probe(void) {
    if (!rccl().ok) return fail(LCX_ERR_COMM, rccl().error);
    return LCX_OK;
}

int lcx_comm_unique_id(void* id_out) {
    if (!id_out) return fail(LCX_ERR_ARG, "lcx_comm_unique_id: null");
    if (!rccl().ok) return fail(LCX_ERR_COMM, rccl().error);
    ncclUniqueId id;
    RCCLCHECK(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == LCX_COMM_ID_BYTES, "LCX_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    memcpy(id_out, &id, sizeof(id));
    return LCX_OK;
}

static int drop_transport(lcx_ctx* h) {
    if (h->tr.kind == 1 && h->tr.comm) {
        (void)hipStreamSynchronize(h->stream);
        (void)rccl().CommDestroy(h->tr.comm);
    }
    h->tr = Transport();
    h->merged_agreed = -1;         // another transport, another group: the ranks agree on the merged pass again
    h->y1_ready = h->w1_ready = h->yk_ready = h->grad_ready = h->early_grad = false;      // nothing of a merged flow survives it
    h->have_direction = false;
    return LCX_OK;
}

int lcx_comm_init(lcx_ctx* h, int nranks, int rank, const void* id_in) {
    NEED_MUT(h);
    if (nranks < 1 || rank < 0 || rank >= nranks || !id_in) return fail(LCX_ERR_ARG, "lcx_comm_init: bad rank / size / id");
    if (!rccl().ok) return fail(LCX_ERR_COMM, rccl().error);
    HIPCHECK(hipStreamSynchronize(h->stream));
    LCXCHECK(drop_transport(h));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCLCHECK(rccl().CommInitRank(&comm, nranks, id, rank));
    h->tr.kind = 1;
    h->tr.comm = comm;
    h->tr.rank = rank;
    h->tr.nranks = nranks;
    h->world = nranks;
    h->exchange = true;          // also for a group of one rank: the caller asked for the multi-rank path
    return LCX_OK;
}

int lcx_set_exchange_hook(lcx_ctx* h, lcx_allreduce_fn fn, void* user) {
    NEED_MUT(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    LCXCHECK(drop_transport(h));
    if (fn) {
        h->tr.kind = 2;
        h->tr.hook = fn;
        h->tr.user = user;
    }
    return LCX_OK;
}

int lcx_comm_selftest(lcx_ctx* h, int rank, int* ok_out, double* seconds_per_allreduce) {
    NEED_MUT(h);
    if (ok_out) *ok_out = 0;
    // (no transport: a property of how the caller set the handle up, the same on every rank - nobody enters a collective)
    if (!h->exchange || h->tr.kind == 0)
        return fail(LCX_ERR_STATE, "lcx_comm_selftest: no transport bound (lcx_comm_init / lcx_set_exchange_hook first)");
    const int64_t n = h->ybuf_main;                       // what every level all-reduces: [Y | tail]
    const int nr = h->world;
    // ---- pre-flight: everything that can fail on ONE rank only happens before the first collective, and its outcome is shared by a
    // one-element all-reduce that every rank enters whatever happened to it - a rank that returned early would leave the others
    // blocked inside the big all-reduce below ----
    std::string local;
    if (rank < 0 || rank >= h->world || (h->tr.kind == 1 && rank != h->tr.rank))
        local = "rank " + std::to_string(rank) + " is not this handle's rank in a world of " + std::to_string(h->world);
    if (local.empty() && nr > 1024) local = "more than 1024 ranks: the rank-identity check sums 16-bit pieces exactly up to 1024 ranks";
    DevTemps tmp;
    unsigned long long* res = nullptr;
    if (local.empty() && tmp.get(&res, 4 * sizeof(unsigned long long)) != LCX_OK) local = g_err;
    if (local.empty() && getenv("LCX_TEST_FAIL_SELFTEST_PREFLIGHT") && atoi(getenv("LCX_TEST_FAIL_SELFTEST_PREFLIGHT")) == rank)
        local = "LCX_TEST_FAIL_SELFTEST_PREFLIGHT";      // test hook: a one-sided local failure
    {
        double flag = local.empty() ? 0.0 : 1.0;
        HIPCHECK(hipStreamSynchronize(h->stream));
        HIPCHECK(hipMemcpyAsync(h->sbuf, &flag, sizeof(double), hipMemcpyHostToDevice, h->stream));
        LCXCHECK(exchange(h, h->sbuf, 1, LCX_F64));
        HIPCHECK(hipMemcpyAsync(&flag, h->sbuf, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipMemsetAsync(h->sbuf, 0, sizeof(double), h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        if (flag != 0.0)
            return fail(local.empty() ? LCX_ERR_COMM : LCX_ERR_ARG,
                        "lcx_comm_selftest: pre-flight failed on " + std::to_string((long long)flag) + " rank(s)" +
                            (local.empty() ? std::string(" (not this one)") : ": " + local));
    }
    unsigned long long host[4] = {0, 0, 0, 0};
    const unsigned grid = (unsigned)std::min<int64_t>(2048, cdiv(n, 256));
    HIPCHECK(hipMemsetAsync(res, 0, 4 * sizeof(unsigned long long), h->stream));
    double secs = 0.0;
    for (int pattern = 0; pattern < 2; ++pattern) {
        if (h->dtype == LCX_F32) hipLaunchKernelGGL((lcx::selftest_fill_kernel<float>), dim3(grid), dim3(256), 0, h->stream, (float*)h->ybuf, n, pattern, rank);
        else hipLaunchKernelGGL((lcx::selftest_fill_kernel<double>), dim3(grid), dim3(256), 0, h->stream, (double*)h->ybuf, n, pattern, rank);
        KCHECK();
        HIPCHECK(hipStreamSynchronize(h->stream));
        const auto t0 = std::chrono::steady_clock::now();
        if (pattern == 1 && h->ypipe > 1) {
            // the pipelined Y exchange hands the transport the library's SECOND stream: the rank-dependent pattern goes that way,
            // ordered by events exactly as y_pass_pipelined orders a chunk - a hook that ignores its stream argument (or a
            // transport that cannot run there) fails here, at first contact, instead of racing with the slot reductions later
            LCXCHECK(ypipe_streams(h));
            HIPCHECK(hipEventRecord(h->ypipe_ev[0], h->stream));
            HIPCHECK(hipStreamWaitEvent(h->comm_stream, h->ypipe_ev[0], 0));
            LCXCHECK(exchange_on(h, h->comm_stream, h->ybuf, n, h->dtype));
            HIPCHECK(hipEventRecord(h->ypipe_ev[16], h->comm_stream));
            HIPCHECK(hipStreamWaitEvent(h->stream, h->ypipe_ev[16], 0));
        } else {
            LCXCHECK(exchange(h, h->ybuf, n, h->dtype));
        }
        HIPCHECK(hipStreamSynchronize(h->stream));
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (h->dtype == LCX_F32) hipLaunchKernelGGL((lcx::selftest_check_kernel<float>), dim3(grid), dim3(256), 0, h->stream, (const float*)h->ybuf, n, pattern, nr, res + 2 * pattern);
        else hipLaunchKernelGGL((lcx::selftest_check_kernel<double>), dim3(grid), dim3(256), 0, h->stream, (const double*)h->ybuf, n, pattern, nr, res + 2 * pattern);
        KCHECK();
    }
    HIPCHECK(hipMemcpyAsync(host, res, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    // every rank learns every rank's verdict through the scalar exchange buffer (doubles): the number of wrong sums, and the
    // four 16-bit pieces a of each result hash with their squares - all ranks hold the same bits iff n * sum(a^2) == (sum a)^2
    // for every piece (Cauchy-Schwarz); with 16-bit pieces both sides stay below 2^53 - exact in doubles - up to 1024 ranks
    constexpr int NSV = 1 + 2 * 4 * 2;
    double sv[NSV] = {0};
    sv[0] = (double)host[0];
    for (int pattern = 0; pattern < 2; ++pattern) {
        const uint64_t hs = host[2 * pattern + 1];
        for (int k = 0; k < 4; ++k) {
            const double part = (double)((hs >> (16 * k)) & 0xFFFFull);
            sv[1 + pattern * 8 + 2 * k] = part;
            sv[2 + pattern * 8 + 2 * k] = part * part;
        }
    }
    HIPCHECK(hipMemcpyAsync(h->sbuf, sv, NSV * sizeof(double), hipMemcpyHostToDevice, h->stream));
    LCXCHECK(exchange(h, h->sbuf, NSV, LCX_F64));
    HIPCHECK(hipMemcpyAsync(sv, h->sbuf, NSV * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    // leave the exchange buffers as lcx_bind_exchange leaves them
    HIPCHECK(hipMemsetAsync(h->ybuf, 0, (size_t)n * h->es, h->stream));
    HIPCHECK(hipMemsetAsync(h->sbuf, 0, NSV * sizeof(double), h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (seconds_per_allreduce) *seconds_per_allreduce = secs / 2.0;
    bool same = true;
    for (int k = 0; k < 8; ++k) same = same && ((double)nr * sv[2 + 2 * k] == sv[1 + 2 * k] * sv[1 + 2 * k]);
    if (sv[0] != 0.0)
        return fail(LCX_ERR_COMM, "lcx_comm_selftest: the all-reduce of " + std::to_string(n) + " elements over " + std::to_string(nr) +
                                      " ranks returned " + std::to_string((long long)sv[0]) + " wrong sums (all ranks together; this rank: " +
                                      std::to_string((long long)host[0]) + ")");
    if (!same)
        return fail(LCX_ERR_COMM, "lcx_comm_selftest: the ranks hold different bits after the same all-reduce (the line-search "
                                  "decisions of lcx_iterate need rank-identical sums)");
    if (ok_out) *ok_out = 1;
    return LCX_OK;
}

int lcx_exchange_info(lcx_ctx* h, int* kind, int* world, int64_t* allreduces_issued) {
    NEED(h);
    if (kind) *kind = h->exchange ? h->tr.kind : -1;
    if (world) *world = h->world;
    if (allreduces_issued) *allreduces_issued = h->n_exchanges;
    return LCX_OK;
}

int lcx_read_state(lcx_ctx* h, int which, double* out) {
    NEED(h);
    WHICH_OK(which);
    if (!out) return fail(LCX_ERR_ARG, "lcx_read_state: null");
    MomentSet& ms = h->set[which];
    if (which == 0 && h->tan_blocks > 0) {
        // update_tangent of the direction in flight is normally summed by the tail of the first trial's evaluation;
        // somebody wants the current solution's state before that
        const int single = !h->exchange;
        const unsigned int seq = single ? ++h->seq_next : 0u;
        if (h->dtype == LCX_F32)
            hipLaunchKernelGGL((tangent_finalize_kernel<float>), dim3(1), dim3(256), 0, h->stream, h->tanpart, h->tan_blocks, h->sbuf, ms.st,
                               ms.hst_dev, seq, single);
        else
            hipLaunchKernelGGL((tangent_finalize_kernel<double>), dim3(1), dim3(256), 0, h->stream, h->tanpart, h->tan_blocks, h->sbuf, ms.st,
                               ms.hst_dev, seq, single);
        KCHECK();
        if (single) ms.seq_expect = seq;
        h->tan_blocks = 0;
    }
    LCXCHECK(wait_published(h, ms));
    const SetState& s = *ms.hst;
    out[LCX_S_TC] = s.tc;
    out[LCX_S_MAX_UJ] = s.max_uj;
    out[LCX_S_INVALID] = (double)s.invalid;
    out[LCX_S_TANGENT] = s.tangent;
    out[LCX_S_SUM_LOG_RJ] = s.sum_log_rj;
    out[5] = out[6] = out[7] = 0.0;
    return LCX_OK;
}

int lcx_set_trial_reuse(lcx_ctx* h, int enable) {
    NEED_MUT(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->reuse_y = enable != 0;
    return LCX_OK;
}

int lcx_set_sample_divisor(lcx_ctx* h, double n_samples) {
    NEED_MUT(h);
    if (!(n_samples >= 1.0)) return fail(LCX_ERR_ARG, "lcx_set_sample_divisor: n_samples must be >= 1");
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->Ndiv = n_samples;
    return LCX_OK;
}

int lcx_set_linear_mode(lcx_ctx* h, int enable) {
    NEED_MUT(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->full_sig = enable != 0;
    return LCX_OK;
}

int lcx_set_exchange(lcx_ctx* h, int enable) {
    NEED_MUT(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->exchange = enable != 0 || h->world > 1;
    return LCX_OK;
}

int lcx_set_world(lcx_ctx* h, int world) {
    NEED_MUT(h);
    if (world < 1) return fail(LCX_ERR_ARG, "lcx_set_world: world must be >= 1");
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->world = world;
    h->exchange = world > 1;
    h->merged_agreed = -1;
    h->y1_ready = h->w1_ready = h->yk_ready = h->grad_ready = h->early_grad = false;
    h->have_direction = false;
    return LCX_OK;
}

int lcx_read_sbuf(lcx_ctx* h, int64_t offset, int64_t count, double* out) {
    NEED(h);
    if (!out || offset < 0 || count < 1 || offset + count > h->sbuf_elems) return fail(LCX_ERR_ARG, "lcx_read_sbuf: bad range");
    HIPCHECK(hipMemcpyAsync(out, h->sbuf + offset, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

int lcx_x_layout(lcx_ctx* h, int* layout) {
    NEED(h);
    if (!layout) return fail(LCX_ERR_ARG, "lcx_x_layout: null");
    *layout = h->panel ? 2 : (h->single_copy ? 1 : 0);
    return LCX_OK;
}

int lcx_f32_gemm(lcx_ctx* h, int* mode) {
    NEED(h);
    if (!mode) return fail(LCX_ERR_ARG, "lcx_f32_gemm: null");
    *mode = h->split ? 1 : 0;
    return LCX_OK;
}

int lcx_bytes_resident(lcx_ctx* h, int64_t* total, int64_t* x_bytes) {
    NEED(h);
    if (total) *total = (int64_t)h->bytes_resident;
    if (x_bytes) *x_bytes = (int64_t)((h->single_copy || h->panel ? 1 : 2) * (size_t)h->Npad * h->ldx * h->es);
    return LCX_OK;
}

int lcx_timing_enable(lcx_ctx* h, int enable) {
    NEED(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    LCXCHECK(timing_collect(h));
    h->timing = enable != 0;
    return LCX_OK;
}

int lcx_timing_sample(lcx_ctx* h, int every) {
    NEED(h);
    if (every < 1) return fail(LCX_ERR_ARG, "lcx_timing_sample: every must be >= 1");
    h->t_every = every;
    for (int k = 0; k < LCX_T_KINDS; ++k) h->t_count[k] = 0;
    return LCX_OK;
}

int lcx_timing_read(lcx_ctx* h, int kind, int64_t* launches, double* total_ms) {
    NEED(h);
    if (kind < 0 || kind >= LCX_T_KINDS) return fail(LCX_ERR_ARG, "lcx_timing_read: kind must be in [0, 7)");
    HIPCHECK(hipStreamSynchronize(h->stream));
    LCXCHECK(timing_collect(h));
    if (launches) *launches = h->t_launch[kind];
    if (total_ms) *total_ms = h->t_ms[kind];
    return LCX_OK;
}

int lcx_timing_reset(lcx_ctx* h) {
    NEED(h);
    HIPCHECK(hipStreamSynchronize(h->stream));
    LCXCHECK(timing_collect(h));
    for (int k = 0; k < LCX_T_KINDS; ++k) { h->t_launch[k] = 0; h->t_pass[k] = 0; h->t_ms[k] = 0.0; h->t_max[k] = 0.0; h->t_skipped[k] = 0; }
    return LCX_OK;
}

int lcx_timing_passes(lcx_ctx* h, int kind, int64_t* passes) {
    NEED(h);
    if (kind < 0 || kind >= LCX_T_KINDS || !passes) return fail(LCX_ERR_ARG, "lcx_timing_passes: bad argument");
    *passes = h->t_pass[kind];
    return LCX_OK;
}

int lcx_geometry(lcx_ctx* h, int64_t* n_pad, int64_t* ldx, int* m_pad, int64_t* info8) {
    NEED(h);
    if (n_pad) *n_pad = h->Npad;
    if (ldx) *ldx = h->ldx;
    if (m_pad) *m_pad = h->Mp;
    if (info8) {
        info8[0] = h->nt_S; info8[1] = h->nt_KW; info8[2] = h->tn_S; info8[3] = h->tn_KW;
        info8[4] = h->nt_bpc; info8[5] = h->tn_bpc; info8[6] = h->pv_grid; info8[7] = h->n_cus;
    }
    return LCX_OK;
}

}  // extern "C"
