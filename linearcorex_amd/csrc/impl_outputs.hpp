// impl_outputs.hpp - Impl<T, CT>: get_covariance (staged row blocks of the rank-m product), predict, invert (entry points:
// lcx_outputs.hip).  Included at the end of engine.hpp.
#pragma once

// get_covariance rows [row0, row0 + nrows) -> out_host (row-major, leading dimension ld_out elements); see CovStage
// Staging shared by get_covariance and predict.  Every resource is created on its own guard: a call that failed half
// way (say a locked-memory limit on the pinned blocks) leaves the stage retryable instead of half built.
template <typename T, int CT>
int Impl<T, CT>::cov_stage(lcx_ctx* h, bool need_op_a, bool need_op_b) {
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

template <typename T, int CT>
void Impl<T, CT>::place_rows(const T* src, int64_t src_ld, T* dst, int64_t dst_ld, int64_t rows, int64_t cols) {
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

template <typename T, int CT>
int Impl<T, CT>::covariance_blocks(lcx_ctx* h, bool syn, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out_host,
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

template <typename T, int CT>
int Impl<T, CT>::covariance_syn(lcx_ctx* h, const void* std_host, int64_t row0, int64_t nrows, void* out_host) {
    MomentSet& s = h->set[0];
    if (!s.xz) return fail(LCX_ERR_STATE, "synergistic covariance needs lcx_syn_moments_b first");
    return covariance_blocks(h, true, 0.0, std_host, row0, nrows, out_host, h->V, nullptr);
}

template <typename T, int CT>
int Impl<T, CT>::covariance_full(lcx_ctx* h, int syn, double eps, const void* std_host, void* out_host, int64_t ld_out, double* ksec) {
    if (syn && !h->set[0].xz) return fail(LCX_ERR_STATE, "synergistic covariance needs lcx_syn_moments_b first");
    return covariance_blocks(h, syn != 0, eps, std_host, 0, h->V, out_host, ld_out, ksec);
}

template <typename T, int CT>
int Impl<T, CT>::covariance(lcx_ctx* h, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out_host) {
    return covariance_blocks(h, false, eps, std_host, row0, nrows, out_host, h->V, nullptr);
}

// theta = (mean, std) of the working dtype -> the stage's device vectors (kind 0: unused)
template <typename T, int CT>
int Impl<T, CT>::stage_theta(lcx_ctx* h, int kind, const void* mean_h, const void* std_h) {
    CovStage& c = *h->cov;
    if (kind == PP_KIND_NONE) return LCX_OK;
    HIPCHECK(hipMemcpyAsync(c.mean_dev, mean_h, sizeof(T) * h->V, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(c.std_dev, std_h, sizeof(T) * h->V, hipMemcpyHostToDevice, h->stream));
    return LCX_OK;
}

// predict (:440-441): out (n_rows x V) = invert(y . X_i Z_j^T), produced in row blocks like get_covariance: the
// rank-Mp product + the inverse marginal map on the device, two pinned staging blocks, the host-side placement of
// block k-1 under the kernel of block k+1 and the copy of block k.
template <typename T, int CT>
int Impl<T, CT>::predict(lcx_ctx* h, const void* y_host, int64_t n_rows, int syn, const void* xz_host, int kind, const void* mean_h,
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
template <typename T, int CT>
int Impl<T, CT>::invert_rows(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean_h, const void* std_h,
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
