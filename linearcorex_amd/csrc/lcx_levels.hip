// lcx_levels.hip - launch geometry, weights, the moment / update levels, lcx_iterate, the synergistic branch, moment readback
// (include/lcx.h): the entry points; the typed work is compiled per precision (levels_typed.hpp).
#include "levels_typed.hpp"

// the handle's dtype -> the typed half (lcx_levels_f32.hip / lcx_levels_f64.hip); nothing of Impl<T, CT> is instantiated here
#undef DISPATCH
#define DISPATCH(h, fn, ...) return (h)->dtype == LCX_F32 ? lcx_f32::fn(__VA_ARGS__) : lcx_f64::fn(__VA_ARGS__)

int lcx_engine_geometry(lcx_ctx* h) { DISPATCH(h, geometry, h); }

extern "C" {

int lcx_set_ws(lcx_ctx* h, const void* w) {
    NEED_MUT(h);
    h->w1_ready = h->y1_ready = false;
    if (!w) return fail(LCX_ERR_ARG, "lcx_set_ws: null");
    DISPATCH(h, set_ws, h, w);
}

static int get_ws_impl(lcx_ctx* h, int which, void* w) { DISPATCH(h, get_ws, h, which, w); }

int lcx_get_ws(lcx_ctx* h, int which, void* w) {
    NEED(h);
    if (!w || which < 0 || which > 1) return fail(LCX_ERR_ARG, "lcx_get_ws: bad argument");
    return get_ws_impl(h, which, w);
}

int lcx_permute_factors(lcx_ctx* h, const int32_t* order) {
    NEED_MUT(h);
    h->w1_ready = h->y1_ready = false;
    if (!order) return fail(LCX_ERR_ARG, "lcx_permute_factors: null");
    for (int j = 0; j < h->M; ++j)
        if (order[j] < 0 || order[j] >= h->M) return fail(LCX_ERR_ARG, "lcx_permute_factors: index out of range");
    DISPATCH(h, permute, h, order);
}

int lcx_moments_a(lcx_ctx* h, int which) { NEED_MUT(h); WHICH_OK(which); DISPATCH(h, moments_a, h, which); }

int lcx_moments_b(lcx_ctx* h, int which, double eps, int quick) { NEED_MUT(h); WHICH_OK(which); DISPATCH(h, moments_b, h, which, eps, quick); }

int lcx_moments_c(lcx_ctx* h, int which) { NEED_MUT(h); WHICH_OK(which); DISPATCH(h, moments_c, h, which); }

static int detail_entry(lcx_ctx* h, int which) { DISPATCH(h, detail, h, which); }

int lcx_moments_detail(lcx_ctx* h, int which) {
    NEED(h);
    WHICH_OK(which);
    LCXCHECK(detail_entry(h, which));
    return exchange(h, h->sbuf + sb_det(h->Mp), h->M + 3, LCX_F64);
}

int lcx_update_a(lcx_ctx* h) { NEED_MUT(h); DISPATCH(h, update_a, h); }

int lcx_update_b(lcx_ctx* h, double eps) { NEED_MUT(h); DISPATCH(h, update_b, h, eps); }

int lcx_update_c(lcx_ctx* h, double eps) { NEED_MUT(h); DISPATCH(h, update_c, h, eps); }

int lcx_update_d(lcx_ctx* h) {
    NEED_MUT(h);
    // With one GPU lcx_update_c already published the tangent; with several ranks its partial sits in sbuf[2]
    // and becomes global with the scalar all-reduce of the first trial (lcx_moments_c stores it).
    h->have_direction = true;
    return LCX_OK;
}

int lcx_make_trial(lcx_ctx* h, double eta) {
    NEED_MUT(h);
    if (!h->have_direction) return fail(LCX_ERR_STATE, "lcx_make_trial before lcx_update_a..d");
    DISPATCH(h, make_trial, h, eta);
}

int lcx_trial_linear_a(lcx_ctx* h, double eta) {
    NEED_MUT(h);
    if (!h->have_direction) return fail(LCX_ERR_STATE, "lcx_trial_linear_a before lcx_update_a..d");
    DISPATCH(h, trial_linear_a, h, eta);
}

int lcx_trial_linear_b(lcx_ctx* h, double eps, double eta) {
    NEED_MUT(h);
    if (!h->have_direction) return fail(LCX_ERR_STATE, "lcx_trial_linear_b before lcx_update_a..d");
    DISPATCH(h, trial_linear_b, h, eps, eta);
}

int lcx_accept_trial(lcx_ctx* h) {
    NEED_MUT(h);
    h->w1_ready = h->y1_ready = false;
    std::swap(h->Wt[0], h->Wt[1]);
    std::swap(h->set[0], h->set[1]);
    h->have_direction = false;
    return LCX_OK;
}

int lcx_iterate(lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out8) {
    NEED(h);
    if (!out8) return fail(LCX_ERR_ARG, "lcx_iterate: null");
    DISPATCH(h, iterate, h, eps, tol, tc_cur, more, out8);
}

int lcx_syn_moments_b(lcx_ctx* h, int which, double yscale) { NEED_MUT(h); WHICH_OK(which); DISPATCH(h, syn_moments_b, h, which, yscale); }

int lcx_syn_moments_c(lcx_ctx* h, int which) { NEED_MUT(h); WHICH_OK(which); DISPATCH(h, syn_moments_c, h, which); }

int lcx_syn_update_a(lcx_ctx* h) { NEED_MUT(h); DISPATCH(h, syn_update_a, h); }

int lcx_syn_update_b(lcx_ctx* h, double eta) { NEED_MUT(h); DISPATCH(h, syn_update_b, h, eta); }

int lcx_rescale_ws(lcx_ctx* h, double e0, double e1) { NEED_MUT(h); h->w1_ready = h->y1_ready = false; DISPATCH(h, rescale, h, e0, e1); }

int lcx_init_scale_ws(lcx_ctx* h) { NEED_MUT(h); h->w1_ready = h->y1_ready = false; DISPATCH(h, init_scale, h); }

int lcx_get_moment(lcx_ctx* h, int which, int key, double eps, void* out) {
    NEED(h);
    WHICH_OK(which);
    if (!out) return fail(LCX_ERR_ARG, "lcx_get_moment: null");
    DISPATCH(h, get_moment, h, which, key, eps, out);
}

int lcx_set_moment(lcx_ctx* h, int which, int key, const void* in) {
    NEED_MUT(h);
    WHICH_OK(which);
    if (!in) return fail(LCX_ERR_ARG, "lcx_set_moment: null");
    DISPATCH(h, set_moment, h, which, key, in);
}

static int split_supported_dispatch(lcx_ctx* h) { DISPATCH(h, split_supported, h); }

int lcx_set_f32_gemm(lcx_ctx* h, int mode) {
    NEED(h);
    if (mode != 0 && mode != 1) return fail(LCX_ERR_ARG, "lcx_set_f32_gemm: mode must be 0 (float32 MFMA) or 1 (bf16 split)");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (mode == 0 || split_supported_dispatch(h) != 1) { h->split = false; return LCX_OK; }
    if (!h->bsp) {
        const int64_t k = h->ldx > h->Npad ? h->ldx : h->Npad;
        const size_t bytes = (size_t)k * (size_t)(h->merged_ok ? 2 * h->Mp : h->Mp) * 6;
        int rc = dev_alloc(&h->bsp, bytes, h->stream);
        if (rc != LCX_OK) { h->bsp = nullptr; return rc; }
        h->bsp_bytes = bytes;
        h->bytes_resident += bytes;
    }
    h->split = true;
    return LCX_OK;
}

int lcx_kernel_name(lcx_ctx* h, int kind, char* buf, int64_t len) {
    if (!h) return fail(LCX_ERR_ARG, "null handle");
    if (kind < 0 || kind > 2 || !buf || len < 16) return fail(LCX_ERR_ARG, "lcx_kernel_name: bad argument");
    DISPATCH(h, kernel_name, h, kind, buf, len);
}

}  // extern "C"
