// impl_levels.hpp - Impl<T, CT>: launch geometry, the X passes and their reductions, the moment / update levels, lcx_iterate, the
// synergistic branch, readback of weights and moments (entry points: lcx_levels.hip; instantiated per precision in lcx_levels_f32/_f64.hip).
// Included at the end of engine.hpp.
#pragma once

// C[z][M][ldc] = sum over split z of opA . B (. rowscale) on gemm_wide; M, N multiples of 64, K of 16
template <typename T, int CT>
template <bool TRANS_A, bool SCALE>
int Impl<T, CT>::wide_gemm(lcx_ctx* h, const T* A, int64_t lda, const T* B, int64_t ldb, const T* scale, T* C, int64_t ldc, int64_t M,
    int64_t N, int64_t K, int S, const int* skip) {
    dim3 grid((unsigned)(N / 64), (unsigned)(M / 64), (unsigned)S);
    hipLaunchKernelGGL((gemm_wide_kernel<T, TRANS_A, SCALE>), grid, dim3(256), 0, h->stream, A, lda, B, ldb, scale, C, ldc, M, K, S, skip);
    KCHECK();
    return LCX_OK;
}

// resident blocks per CU of a kernel at a given block size / dynamic LDS
template <typename T, int CT>
template <typename F>
int Impl<T, CT>::blocks_per_cu(F* f, int threads, size_t lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)f, threads, lds) != hipSuccess || n < 1) n = 1;
    return n;
}

template <typename T, int CT>
int Impl<T, CT>::env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Split of the contraction.  These launches are HBM-bound and every block lives for the whole
// launch, so what matters is how full the last "round" of resident blocks is: 1.03 rounds cost
// almost 2 (measured: 632 blocks on 512 slots 102 us vs 474 blocks 79 us).  Pick the split that
// fills whole rounds best, with a small penalty per extra split (partial tiles to write + sum).
template <typename T, int CT>
int Impl<T, CT>::single_round_split(int64_t tiles, int64_t capacity_blocks, int64_t kunits, int kw, int cap) {
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

template <typename T, int CT>
int Impl<T, CT>::geometry(lcx_ctx* h) {
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

template <typename T, int CT>
int Impl<T, CT>::geometry_tuned(lcx_ctx* h) {
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
template <typename T, int CT>
int Impl<T, CT>::gram(lcx_ctx* h, const T* A, int64_t K, const T* scale, int S, const int* skip, T* dst) {
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
template <typename T, int CT>
int Impl<T, CT>::nt_pass(lcx_ctx* h, const T* B, const int* skip, T* dst, int64_t r0, int64_t rows) {
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
template <typename T, int CT>
int Impl<T, CT>::nt_reduce(lcx_ctx* h, const int* skip, T* also, int64_t e0, int64_t n, hipStream_t st) {
    if (!st) st = h->stream;
    const int64_t ntot = h->Npad * Mp;
    const bool wide = h->nt_S >= WIDE_SPLITS && cdiv(ntot, 32) < (1 << 20);
    if (wide) {
        hipLaunchKernelGGL((reduce_partials_wide_kernel<T, T>), dim3((unsigned)cdiv(n, 32)), dim3(256), 0,
                           st, P<T>(h->ypart) + e0, h->nt_S, n, ntot, P<T>(h->ybuf) + e0, skip, also ? also + e0 : also);
        KCHECK();
    } else if (h->nt_S > 1) {
        hipLaunchKernelGGL((reduce_partials_kernel<T, T>), dim3((unsigned)(cdiv(n, 256) < 1024 ? cdiv(n, 256) : 1024)), dim3(256), 0,
                           st, P<T>(h->ypart) + e0, h->nt_S, n, ntot, P<T>(h->ybuf) + e0, skip, also ? also + e0 : also);
        KCHECK();
    }
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::nt_big(lcx_ctx* h, const void* Bv, const int* skip, bool with_bj, T* also) {
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
template <typename T, int CT>
int Impl<T, CT>::ypipe_init(lcx_ctx* h) {
    return ypipe_streams(h);
}

// B = the weights of the evaluated set (lcx_moments_a: tail = W.W^T partial) or, with_bj, the gradient (lcx_update_b: tail = the
// Bj partial sums of grad_kernel, :302)
template <typename T, int CT>
int Impl<T, CT>::y_pass_pipelined(lcx_ctx* h, const T* w, bool with_bj) {
    LCXCHECK(ypipe_init(h));
    // chunk boundaries in units of 64 rows, a function of the replicated n_samples alone: every rank must all-reduce the same
    // chunks, whatever kernels its own shard width selected (64 is a multiple of every row tile of the wave-split kernels)
    const int64_t tile = 64;
    bool chunk_pass = false;
    if constexpr (!WIDE) chunk_pass = !(h->panel || h->single_copy || h->nt_ct);
    const int64_t tiles = h->Npad / tile;
    const int C = (int)(h->ypipe < tiles ? h->ypipe : tiles);
    // A row chunk of the pass is a launch of tiles / C x nt_S blocks: worth it only while that still fills one round of resident
    // blocks - the whole launch is sized to exactly that (single_round_split), so a quarter of it leaves three quarters of the chip
    // idle and the pass, HBM-bound, takes about as long per chunk as in one piece (config 2: 157 row tiles x 3 splits = 471 blocks
    // on 512 slots).  Otherwise the pass stays one launch and only the slot reductions and all-reduces go out in chunks
    // ("chunks:n:pass" forces the per-chunk pass: tests).
    if constexpr (!WIDE) {
        const int64_t pass_tiles = h->Npad / (16 * Geo<T, CT>::TN_RT);
        if (chunk_pass && !h->ypipe_force_pass && (pass_tiles / C) * (int64_t)h->nt_S < (int64_t)h->n_cus * h->nt_bpc) chunk_pass = false;
    }
    // the tail first: it rides in the last chunk's all-reduce
    if (with_bj) {
        hipLaunchKernelGGL((reduce_y_bj_kernel<T, false>), dim3(Mp), dim3(PV_THREADS), 0, h->stream, P<T>(h->ypart), 1, h->Npad * Mp,
                           P<T>(h->ybuf), 0, h->bjpart, h->pv_grid, Mp, P<T>(h->ybuf) + h->Npad * Mp);
        KCHECK();
    } else {
        LCXCHECK(gram_w(h, w));
    }
    const int site = with_bj ? LCX_T_AR_DIR : LCX_T_AR_Y;
    T* dst = h->nt_S > 1 ? P<T>(h->ypart) : P<T>(h->ybuf);
    if constexpr (!WIDE) {
        // in-launch chunk signalling: the wave-split kernels
        if (h->ypipe_signal && !(h->panel || h->single_copy || h->nt_ct)) {
            LCXCHECK(ypipe_signals(h));
            constexpr int RTT = Geo<T, CT>::TN_RT;          // row tile of the pass: 16 * RTT rows (a divisor of the 64-row chunk unit)
            ChunkSig sg;
            sg.chunk_cnt = h->sig_counters;
            for (int c = 0; c < SIG_MAX_CHUNKS; ++c) sg.flag[c] = h->sig_flag[c];
            for (int c = 0; c <= C; ++c) sg.tile_begin[c] = (int)(tiles * c / C * tile / (16 * RTT));
            sg.nchunks = C;
            sg.epoch = ++h->sig_epoch;
            const unsigned nblocks = (unsigned)(h->Npad / (16 * RTT)) * (unsigned)h->nt_S;
            TimingPair tp;
            LCXCHECK(timing_begin(h, 0, &tp));
            bool launched = false;
            if (h->f64_4x4) {
                if constexpr (sizeof(T) == 8 && CT <= 2) {
                    constexpr int U4 = 4;
                    if (h->nt_KW == 2) {
                        const size_t lds = Tn4Lds<CT, RTT, 2, U4>::bytes;
                        LCXCHECK(allow_lds(gemm_tn4_sig_kernel<CT, RTT, 2, U4>, lds));
                        hipLaunchKernelGGL((gemm_tn4_sig_kernel<CT, RTT, 2, U4>), dim3(nblocks), dim3(128), lds, h->stream, P<double>(h->XT), h->Npad,
                                           (const double*)w, (double*)dst, h->Npad, (int)(h->ldx / 16), h->nt_S, sg);
                    } else {
                        const size_t lds = Tn4Lds<CT, RTT, 4, U4>::bytes;
                        LCXCHECK(allow_lds(gemm_tn4_sig_kernel<CT, RTT, 4, U4>, lds));
                        hipLaunchKernelGGL((gemm_tn4_sig_kernel<CT, RTT, 4, U4>), dim3(nblocks), dim3(256), lds, h->stream, P<double>(h->XT), h->Npad,
                                           (const double*)w, (double*)dst, h->Npad, (int)(h->ldx / 16), h->nt_S, sg);
                    }
                    launched = true;
                }
            } else {
                int kw = h->nt_KW;
                if (kw > 4 && CT >= 16) kw = 4;
                const size_t lds = (size_t)kw * 16 * RTT * 16 * CT * sizeof(T);
#define LCX_SIG_LAUNCH(KWV)                                                                                                              \
    {                                                                                                                                    \
        LCXCHECK(allow_lds(gemm_tn_sig_kernel<T, CT, RTT, KWV>, lds));                                                                    \
        hipLaunchKernelGGL((gemm_tn_sig_kernel<T, CT, RTT, KWV>), dim3(nblocks), dim3(64 * KWV), lds, h->stream, P<T>(h->XT), h->Npad,     \
                           (int64_t)(16 * RTT), w, dst, h->Npad, (int)(h->ldx / 16), h->nt_S, sg);                                         \
    }
                switch (kw) {
                    case 1: LCX_SIG_LAUNCH(1); break;
                    case 2: LCX_SIG_LAUNCH(2); break;
                    case 4: LCX_SIG_LAUNCH(4); break;
                    default: LCX_SIG_LAUNCH(MaxKw<CT>::v); break;
                }
#undef LCX_SIG_LAUNCH
                launched = true;
            }
            if (launched) {
                KCHECK();
                LCXCHECK(timing_end(h, 0, &tp));
                unsigned int* err = h->sig_counters + SIG_MAX_CHUNKS;
                for (int c = 0; c < C; ++c) {
                    const int64_t r0 = tiles * c / C * tile, r1 = tiles * (c + 1) / C * tile;
                    if (h->ypipe_poll) {
                        hipLaunchKernelGGL(poll_signal_kernel, dim3(1), dim3(1), 0, h->comm_stream, h->sig_flag[c], sg.epoch, err);
                        KCHECK();
                    } else {
                        HIPCHECK(hipStreamWaitValue32(h->comm_stream, h->sig_flag[c], sg.epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    }
                    // the chunk's slots -> the summed Y: the reduction of the unpipelined path on this range (same kernel, same bits), here
                    LCXCHECK(nt_reduce(h, nullptr, (T*)nullptr, r0 * Mp, (r1 - r0) * Mp, h->comm_stream));
                    const int64_t count = (r1 - r0) * Mp + (c == C - 1 ? (int64_t)Mp * Mp : 0);
                    LCXCHECK(exchange_site_on(h, h->comm_stream, site, P<T>(h->ybuf) + r0 * Mp, count, DT));
                }
                HIPCHECK(hipEventRecord(h->ypipe_ev[16], h->comm_stream));
                HIPCHECK(hipStreamWaitEvent(h->stream, h->ypipe_ev[16], 0));
                return LCX_OK;
            }
        }
    }
    TimingPair tp;
    tp.kind = -1;
    // (per-chunk launches of the pass: the pair would also span the slot reductions between them - the pass is counted, not timed)
    if (!chunk_pass) LCXCHECK(timing_begin(h, 0, &tp));
    else if (h->timing) h->t_pass[0] += 1;
    if (!chunk_pass) {
        LCXCHECK(nt_pass(h, w, nullptr, dst));
        LCXCHECK(timing_end(h, 0, &tp));
    }
    for (int c = 0; c < C; ++c) {
        const int64_t r0 = tiles * c / C * tile, r1 = tiles * (c + 1) / C * tile;
        if (chunk_pass) LCXCHECK(nt_pass(h, w, nullptr, dst, r0, r1 - r0));
        LCXCHECK(nt_reduce(h, nullptr, (T*)nullptr, r0 * Mp, (r1 - r0) * Mp));
        HIPCHECK(hipEventRecord(h->ypipe_ev[c], h->stream));
        HIPCHECK(hipStreamWaitEvent(h->comm_stream, h->ypipe_ev[c], 0));
        const int64_t count = (r1 - r0) * Mp + (c == C - 1 ? (int64_t)Mp * Mp : 0);
        LCXCHECK(exchange_site_on(h, h->comm_stream, site, P<T>(h->ybuf) + r0 * Mp, count, DT));
    }
    HIPCHECK(hipEventRecord(h->ypipe_ev[16], h->comm_stream));
    HIPCHECK(hipStreamWaitEvent(h->stream, h->ypipe_ev[16], 0));
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::tn_big(lcx_ctx* h, const int* skip) {
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
template <typename T, int CT>
int Impl<T, CT>::gram_w(lcx_ctx* h, const T* w) {
    LCXCHECK(gram(h, w, h->ldx, nullptr, h->gv_S, nullptr, P<T>(h->gpartw)));
    if (h->exchange) {
        hipLaunchKernelGGL((reduce_wide_kernel<T, T>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                           P<T>(h->gpartw), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp,
                           P<T>(h->ybuf) + h->Npad * Mp, (const int*)nullptr);
        KCHECK();
    }
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::moments_a(lcx_ctx* h, int which) {
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
    return exchange_site(h, LCX_T_AR_Y, h->ybuf, h->ybuf_main, DT);           // L1 of SURVEY 8e: [Y_partial | W.W^T partial]
}

// per-factor moments; ysrc != null: first form the Y^T.Y partials of that Y
template <typename T, int CT>
int Impl<T, CT>::small(lcx_ctx* h, int which, double eps, int quick, const T* ysrc) {
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
template <typename T, int CT>
int Impl<T, CT>::gram_pair(lcx_ctx* h, const T* w, const T* y) {
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

template <typename T, int CT>
int Impl<T, CT>::gram_pair_tuned(lcx_ctx* h, const T* w, const T* y) {
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

template <typename T, int CT>
int Impl<T, CT>::epilogue(lcx_ctx* h, int which, double eps, bool linear, double eta) {
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
    return exchange_site(h, LCX_T_AR_S, h->sbuf, SB_H + Mp * Mp, LCX_F64);    // L2 + L3 + L5: TC sums, pending tangent, H of this set
}

template <typename T, int CT>
int Impl<T, CT>::moments_b(lcx_ctx* h, int which, double eps, int quick) {
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
template <typename T, int CT>
int Impl<T, CT>::trial_linear_a(lcx_ctx* h, double eta) {
    const int64_t n1 = h->V * Mp, n2 = h->Npad * Mp;
    hipLaunchKernelGGL((axpy2_kernel<T>), dim3((unsigned)(cdiv(n1 + n2, 256) < 2048 ? cdiv(n1 + n2, 256) : 2048)), dim3(256), 0,
                       h->stream, P<T>(h->Wt[0]), P<T>(h->update), P<T>(h->Wt[1]), n1,
                       P<T>(h->set[0].Y), P<T>(h->ydir), P<T>(h->set[1].Y), n2, (T)eta);
    KCHECK();
    LCXCHECK(gram_pair(h, P<T>(h->Wt[1]), P<T>(h->set[1].Y)));
    return exchange_site(h, LCX_T_AR_SMALL, P<T>(h->ybuf) + h->Npad * Mp, (int64_t)Mp * Mp, DT);      // W'.W'^T partial
}

// b: (W.W^T tail global) uj, flag, D' = D + eta*D(update), rho ... TC partial sums
template <typename T, int CT>
int Impl<T, CT>::trial_linear_b(lcx_ctx* h, double eps, double eta) {
    LCXCHECK(small(h, 1, eps, 1, nullptr));
    return epilogue(h, 1, eps, true, eta);
}

template <typename T, int CT>
int Impl<T, CT>::moments_c(lcx_ctx* h, int which) {
    if (!h->exchange) return LCX_OK;             // the epilogue already published
    MomentSet& s = h->set[which];
    const unsigned int seq = ++h->seq_next;
    hipLaunchKernelGGL((tc_final_kernel<T>), dim3(1), dim3(1), 0, h->stream, h->sbuf, s.st, s.hst_dev, seq);
    KCHECK();
    s.seq_expect = seq;
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::update_a(lcx_ctx* h) {
    MomentSet& s = h->set[0];
    LCXCHECK(gram(h, P<T>(s.rir), h->ldx, P<T>(s.hscale), h->gv_S, nullptr, P<T>(h->gpart)));
    hipLaunchKernelGGL((reduce_wide_kernel<T, double>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                       P<T>(h->gpart), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp, h->sbuf + SB_H, (const int*)nullptr);
    KCHECK();
    return exchange_site(h, LCX_T_AR_SMALL, h->sbuf + SB_H, (int64_t)Mp * Mp, LCX_F64);
}

// grad (:296-300) and the per-block Bj partials (:302) of set `which`, from its moments and the H its evaluation left in sbuf
template <typename T, int CT>
int Impl<T, CT>::launch_grad(lcx_ctx* h, int which) {
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

template <typename T, int CT>
int Impl<T, CT>::update_b(lcx_ctx* h, double eps) {
    (void)eps;
    if (h->spec_dirty) {                            // an abandoned speculation overwrote the H of set 0 in sbuf
        if (!self_contained(h)) return fail(LCX_ERR_STATE, "abandoned lcx_iterate speculation while the caller owns the exchange");
        LCXCHECK(update_a(h));
        h->spec_dirty = false;
    }
    MomentSet& s = h->set[0];
    LCXCHECK(agree_on_merged(h));
    const bool merged = use_merged(h);
    if (h->grad_ready) h->grad_ready = false;       // lcx_iterate already computed it behind the accepted trial's evaluation
    else LCXCHECK(launch_grad(h, 0));
    if (!merged) {
        if (h->exchange && h->ypipe > 1 && h->tr.kind != 0) return y_pass_pipelined(h, P<T>(h->grad), true);
        LCXCHECK(nt_big(h, P<T>(h->grad), nullptr, true));
        return exchange_site(h, LCX_T_AR_DIR, h->ybuf, h->ybuf_main, DT);       // L4: [Y_g partial | Bj partial]
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
            LCXCHECK(exchange_site(h, LCX_T_AR_SMALL, P<T>(h->ybuf) + n, Mp, DT));
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
            LCXCHECK(exchange_site(h, LCX_T_AR_DIR, h->ybuf, h->ybuf_elems, DT));
        }
        h->w1_ready = h->y1_ready = true;
    }
    return LCX_OK;
}

// the merged pass needs: its buffers, the Y-space tangent, and - with several ranks - the exchange inside the library (a caller
// that all-reduces the buffers itself between the levels does not know about the Bj exchange in front of the pass)
// One all-reduce of one flag, once per transport: does EVERY rank's shard have a merged form?  (Uneven shards: a rank of 64 variables
// next to one of 1333 - tests/test_thread_ranks_gpu.py caught ranks issuing different collectives.)  Every rank gets here at the
// same point of the same program: the first lcx_update_b.
template <typename T, int CT>
int Impl<T, CT>::agree_on_merged(lcx_ctx* h) {
    if (h->merged_agreed >= 0) return LCX_OK;
    if (!h->exchange || h->tr.kind == 0 || h->world <= 1) { h->merged_agreed = 1; return LCX_OK; }
    double flag = h->merged_ok ? 1.0 : 0.0;
    HIPCHECK(hipMemcpyAsync(h->sbuf + 3, &flag, sizeof(double), hipMemcpyHostToDevice, h->stream));
    LCXCHECK(exchange(h, h->sbuf + 3, 1, LCX_F64));
    HIPCHECK(hipMemcpyAsync(&flag, h->sbuf + 3, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipMemsetAsync(h->sbuf + 3, 0, sizeof(double), h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->merged_agreed = flag == (double)h->world ? 1 : 0;
    return LCX_OK;
}
template <typename T, int CT>
bool Impl<T, CT>::use_merged(const lcx_ctx* h) {
    // == 1: "not asked yet" (-1) is not an agreement - only lcx_update_b asks, and whatever resets the answer (lcx_set_world, another
    // transport) also drops what an earlier merged pass left behind (y1_ready / w1_ready / a gradient computed ahead)
    return h->merged_ok && h->merged_agreed == 1 && (!h->exchange || h->tr.kind != 0) && !h->full_sig && h->gw != nullptr;
}

template <typename T, int CT>
int Impl<T, CT>::update_grid(const lcx_ctx* h) {
    return (int)(cdiv(h->V * Mp, PV_THREADS) < 1536 ? cdiv(h->V * Mp, PV_THREADS) : 1536);
}

template <typename T, int CT>
int Impl<T, CT>::update_c(lcx_ctx* h, double eps) {
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

template <typename T, int CT>
int Impl<T, CT>::make_trial(lcx_ctx* h, double eta) {
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
template <typename T, int CT>
int Impl<T, CT>::trial_by_linearity(lcx_ctx* h, double eta) {
    const int64_t n1 = h->V * Mp, n2 = h->Npad * Mp;
    hipLaunchKernelGGL((axpy2_kernel<T>), dim3((unsigned)(cdiv(n1 + n2, 256) < 2048 ? cdiv(n1 + n2, 256) : 2048)), dim3(256), 0,
                       h->stream, P<T>(h->Wt[0]), P<T>(h->update), P<T>(h->Wt[1]), n1,
                       P<T>(h->set[0].Y), P<T>(h->ydir), P<T>(h->ybuf), n2, (T)eta);
    KCHECK();
    if (h->exchange) {
        // Y and X.update^T are already sums over all ranks; only W'.W'^T of the trial is still per shard
        LCXCHECK(gram_w(h, P<T>(h->Wt[1])));
        LCXCHECK(exchange_site(h, LCX_T_AR_SMALL, P<T>(h->ybuf) + h->Npad * Mp, (int64_t)Mp * Mp, DT));
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
template <typename T, int CT>
int Impl<T, CT>::evaluate_trial(lcx_ctx* h, double eps) {
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

template <typename T, int CT>
int Impl<T, CT>::direction_and_trial(lcx_ctx* h, double eps) {
    LCXCHECK(update_b(h, eps));                  // grad (:296-300), Y_g = X.grad^T (:210), Bj (:302)
    LCXCHECK(update_c(h, eps));                  // update (:303), update_tangent partials (:305), ws + update
    h->have_direction = true;
    LCXCHECK(make_trial(h, 1.0));                // :320 at eta = 1 (update_kernel wrote it already)
    return evaluate_trial(h, eps);
}

template <typename T, int CT>
int Impl<T, CT>::iterate(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out) {
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

template <typename T, int CT>
int Impl<T, CT>::iterate_body(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out) {
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

template <typename T, int CT>
int Impl<T, CT>::rescale(lcx_ctx* h, double e0, double e1) {
    const int64_t n = h->V * Mp;
    hipLaunchKernelGGL((rescale_kernel<T>), dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0,
                       h->stream, P<T>(h->Wt[0]), n, Mp, h->set[0].uj, h->set[0].wmag, e0, e1);
    KCHECK();
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::init_scale(lcx_ctx* h) {
    LCXCHECK(small(h, 0, 0.0, 0, P<T>(h->ybuf)));
    const int64_t n = h->V * Mp;
    hipLaunchKernelGGL((init_scale_kernel<T>), dim3((unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0,
                       h->stream, P<T>(h->Wt[0]), n, Mp, h->M, h->set[0].uj);
    KCHECK();
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::permute(lcx_ctx* h, const int32_t* order) {
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
template <typename T, int CT>
int Impl<T, CT>::detail(lcx_ctx* h, int which, T* mi_o, T* xz_o, T* x2y_o) {
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
template <typename T, int CT>
int Impl<T, CT>::syn_alloc(lcx_ctx* h) {
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
template <typename T, int CT>
int Impl<T, CT>::syn_moments_b(lcx_ctx* h, int which, double yscale) {
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

template <typename T, int CT>
int Impl<T, CT>::syn_moments_c(lcx_ctx* h, int which) {
    MomentSet& s = h->set[which];
    const unsigned int seq = ++h->seq_next;
    hipLaunchKernelGGL((syn_tc_kernel<T>), dim3(1), dim3(1), 0, h->stream, h->sbuf + sb_det(Mp), h->M, s.st, s.hst_dev, seq);
    KCHECK();
    s.seq_expect = seq;
    return LCX_OK;
}

// H partial (:378) -> sbuf[0 .. Mp^2)
template <typename T, int CT>
int Impl<T, CT>::syn_update_a(lcx_ctx* h) {
    MomentSet& s = h->set[0];
    if (!s.xz) return fail(LCX_ERR_STATE, "lcx_syn_update_a before lcx_syn_moments_b");
    LCXCHECK(gram(h, P<T>(s.xz), h->ldx, P<T>(s.hscale), h->gv_S, nullptr, P<T>(h->gpart)));
    hipLaunchKernelGGL((reduce_wide_kernel<T, double>), dim3(cdiv(Mp * Mp, 32)), dim3(256), 0, h->stream,
                       P<T>(h->gpart), h->gv_S, (int64_t)Mp * Mp, (int64_t)Mp * Mp, h->sbuf + SB_H, (const int*)nullptr);
    KCHECK();
    return exchange(h, h->sbuf + SB_H, (int64_t)Mp * Mp, LCX_F64);
}

// ws' = (1-eta) ws + eta (R - H ws) (:380-382) -> set 1
template <typename T, int CT>
int Impl<T, CT>::syn_update_b(lcx_ctx* h, double eta) {
    MomentSet& s = h->set[0];
    const size_t lds = ((size_t)(OpInLds<Mp>::v ? Mp * (Mp + 1) : 0) + (size_t)VPB * Mp) * sizeof(T);
    LCXCHECK(allow_lds(syn_update_kernel<T, Mp>, lds));
    hipLaunchKernelGGL((syn_update_kernel<T, Mp>), dim3(h->pv_grid), dim3(NTV), lds, h->stream, P<T>(h->Wt[0]), P<T>(s.xz),
                       P<T>(s.hscale), h->sbuf + SB_H, h->V, (T)eta, P<T>(h->Wt[1]));
    KCHECK();
    return LCX_OK;
}

// [Vp][Mp] device -> (m, V) or (V, m) host
template <typename T, int CT>
int Impl<T, CT>::fetch_mv(lcx_ctx* h, const T* dev, T* host, bool as_m_by_v) {
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

template <typename T, int CT>
int Impl<T, CT>::fetch_v(lcx_ctx* h, const T* dev, T* host) {
    HIPCHECK(hipMemcpyAsync(host, dev, (size_t)h->V * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::fetch_small(lcx_ctx* h, const double* dev, int rows, int cols, T* host) {
    std::vector<double> tmp((size_t)Mp * Mp);
    HIPCHECK(hipMemcpyAsync(tmp.data(), dev, sizeof(double) * (rows == 1 ? Mp : Mp * Mp), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (int a = 0; a < rows; ++a)
        for (int b = 0; b < cols; ++b) host[a * cols + b] = (T)tmp[(rows == 1 ? 0 : a * Mp) + b];
    return LCX_OK;
}

template <typename T, int CT>
int Impl<T, CT>::get_moment(lcx_ctx* h, int which, int key, double eps, void* out) {
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

template <typename T, int CT>
int Impl<T, CT>::set_moment(lcx_ctx* h, int which, int key, const void* in) {
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

template <typename T, int CT>
int Impl<T, CT>::set_ws(lcx_ctx* h, const void* w_host) {
    const T* w = reinterpret_cast<const T*>(w_host);
    std::vector<T> tmp((size_t)h->ldx * Mp, (T)0);
    for (int j = 0; j < h->M; ++j)
        for (int64_t v = 0; v < h->V; ++v) tmp[v * Mp + j] = w[(int64_t)j * h->V + v];
    HIPCHECK(hipMemcpyAsync(h->Wt[0], tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

// split mode (gemm_split_kernels.hpp) needs the panel-major copy, float32 and 32 / 64 / 128 padded factors
template <typename T, int CT>
int Impl<T, CT>::split_supported(lcx_ctx* h) {
    if constexpr (!WIDE && split_capable<T, CT>()) {
        return h->panel ? 1 : 0;
    } else {
        (void)h;
        return 0;
    }
}

// Name of the kernel instantiation behind the two X-streaming passes, as rocprofv3 prints it (both
// passes run the same function: X.B^T contracts over the rows of the transposed copy).
template <typename T, int CT>
int Impl<T, CT>::kernel_name(lcx_ctx* h, int kind, char* buf, int64_t len) {
    if constexpr (WIDE) {
        if (kind == 2) buf[0] = 0;
        else snprintf(buf, (size_t)len, "lcx::gemm_wide_kernel<%s, %s, false>", sizeof(T) == 8 ? "double" : "float", kind == 0 ? "false" : "true");
        return LCX_OK;
    } else {
        return kernel_name_tuned(h, kind, buf, len);
    }
}

template <typename T, int CT>
int Impl<T, CT>::kernel_name_tuned(lcx_ctx* h, int kind, char* buf, int64_t len) {
    if (h->split) {
        if (kind == 2 && (!h->merged_ok || h->merged_agreed == 0)) { buf[0] = 0; return LCX_OK; }   // the group agreed not to take it
        const int ct = kind == 2 ? 2 * CT : CT;
        snprintf(buf, (size_t)len, "lcx::gemm_split_kernel<%d, %d, 6, %s, true, false, 2, %d, %d>", ct, ct >= 4 ? 8 : 4, kind == 1 ? "true" : "false",
                 ct <= 4 ? 2 : 1, ct == 8 ? 2 : 1);
        return LCX_OK;
    }
    if (kind == 2) {
        if (!h->merged_ok || h->merged_agreed == 0) { buf[0] = 0; return LCX_OK; }
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
