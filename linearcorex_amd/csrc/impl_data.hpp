// impl_data.hpp - Impl<T, CT>: making a shard resident (upload, the three preprocess passes, the panel staging, the on-device generator),
// reading it back, transform of new rows (entry points: lcx_data.hip).  Included at the end of engine.hpp.
#pragma once

template <typename T, int CT>
int Impl<T, CT>::make_xt(lcx_ctx* h) {
    if (h->single_copy || h->panel) { HIPCHECK(hipStreamSynchronize(h->stream)); return LCX_OK; }
    dim3 grid((unsigned)(h->ldx / 64), (unsigned)(h->Npad / 64));
    hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, h->stream, P<T>(h->X), h->ldx, P<T>(h->XT), h->Npad);
    KCHECK();
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

// y[rows_pad][Mp] = xd[rows_pad][ldx] . W^T for a staged block of new rows (transform, :386-395).  Up to 128 padded
// factors: the row-streaming kernel gemm_nt.  256: its register tile does not fit, so the block is transposed and runs
// through the column-streaming kernel like the resident passes do.
template <typename T, int CT>
int Impl<T, CT>::project_block(lcx_ctx* h, DevTemps& tmps, T* xd, int64_t rows_pad, T* yd, T** xt_io) {
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

template <typename T, int CT>
int Impl<T, CT>::project(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, void* out_host) {
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

// ---- preprocess on device (:397-429): stats + impute + standardise / tail squash, in place ----
// on a ROW-MAJOR view X[Npad][ldx] of V columns: the resident shard itself, or - panel layout - a staged block of its columns
// (every step of :397-429 is per column, so blocks of columns are preprocessed independently).
// mean_io / std_io: host arrays of T (V entries: the caller offsets them to the view's first column); nobs_out: int64 (may be null);
// xt_view: [ldx][Npad] to leave the transposed copy in when kind is 'empirical' (nullptr: a temporary for the sort)
template <typename T, int CT>
int Impl<T, CT>::preprocess_resident(lcx_ctx* h, int kind, int has_missing, double sentinel, int fit, void* mean_io,
    void* std_io, int64_t* nobs_out, double* maxabs_out) {
    return preprocess_view(h, P<T>(h->X), h->V, h->ldx, kind, has_missing, sentinel, fit, mean_io, std_io, nobs_out, maxabs_out,
                           h->single_copy ? (T*)nullptr : P<T>(h->XT));
}

template <typename T, int CT>
int Impl<T, CT>::preprocess_view(lcx_ctx* h, T* X, const int64_t V, const int64_t ldx, int kind, int has_missing, double sentinel, int fit,
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
template <typename T, int CT>
int64_t Impl<T, CT>::panel_block_cols(const lcx_ctx* h) {
    int64_t w = (((int64_t)1 << 28) / h->Npad) / 64 * 64;
    const int forced = env_int("LCX_PANEL_BLOCK_COLS", 0);          // test hook: several blocks at small sizes
    if (forced > 0) w = (int64_t)forced / 64 * 64;
    if (w < 64) w = 64;
    return w > h->ldx ? h->ldx : w;
}

template <typename T, int CT>
template <typename F>
int Impl<T, CT>::panel_fill(lcx_ctx* h, F fill) {
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

template <typename T, int CT>
int Impl<T, CT>::upload_x(lcx_ctx* h, const void* x, int64_t ld) {
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

template <typename T, int CT>
int Impl<T, CT>::download_x(lcx_ctx* h, void* x, int64_t ld) {
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

template <typename T, int CT>
int Impl<T, CT>::upload_preprocess(lcx_ctx* h, const void* x, int64_t ld, int kind, int has_missing, double sentinel, int fit,
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
template <typename T, int CT>
int Impl<T, CT>::project_raw(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean_h,
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

template <typename T, int CT>
int Impl<T, CT>::generate(lcx_ctx* h, uint64_t seed, int kind, int n_groups, int64_t col_offset) {
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
