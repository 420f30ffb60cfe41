// lcx_data.hip - making a shard resident (upload + preprocess, the on-device generator), reading it back, transform of new rows
// (include/lcx.h)
#include "engine.hpp"

extern "C" {

int lcx_upload_x(lcx_ctx* h, const void* x, int64_t ld) {
    NEED_MUT(h);
    if (!x || ld < h->V) return fail(LCX_ERR_ARG, "lcx_upload_x: bad leading dimension");
    DISPATCH(h, upload_x, h, x, ld);
}

int lcx_upload_preprocess(lcx_ctx* h, const void* x, int64_t ld, int kind, int has_missing, double missing, int fit, void* mean_io,
                          void* std_io, int64_t* n_obs_out, double* max_abs_out) {
    NEED_MUT(h);
    if (!x || ld < h->V) return fail(LCX_ERR_ARG, "lcx_upload_preprocess: bad leading dimension");
    if (kind < 0 || kind > 3) return fail(LCX_ERR_ARG, "lcx_upload_preprocess: kind must be 0 (none), 1 (standard), 2 (outliers) or 3 (empirical)");
    DISPATCH(h, upload_preprocess, h, x, ld, kind, has_missing, missing, fit, mean_io, std_io, n_obs_out, max_abs_out);
}

int lcx_download_x(lcx_ctx* h, void* x, int64_t ld) {
    NEED(h);
    if (!x || ld < h->V) return fail(LCX_ERR_ARG, "lcx_download_x: bad leading dimension");
    DISPATCH(h, download_x, h, x, ld);
}

int lcx_generate_x(lcx_ctx* h, uint64_t seed, int kind, int n_groups, int64_t col_offset) {
    NEED_MUT(h);
    DISPATCH(h, generate, h, seed, kind, n_groups, col_offset);
}

int lcx_project(lcx_ctx* h, const void* x, int64_t n_rows, int64_t ld, void* out) {
    NEED(h);
    if (!x || !out || n_rows < 1 || ld < h->V) return fail(LCX_ERR_ARG, "lcx_project: bad argument");
    DISPATCH(h, project, h, x, n_rows, ld, out);
}

int lcx_project_raw(lcx_ctx* h, const void* x, int64_t n_rows, int64_t ld, int kind, const void* mean, const void* stdv, void* out) {
    NEED(h);
    if (!x || !out || n_rows < 1 || ld < h->V) return fail(LCX_ERR_ARG, "lcx_project_raw: bad argument");
    if (kind < 0 || kind > 2 || (kind != 0 && (!mean || !stdv))) return fail(LCX_ERR_ARG, "lcx_project_raw: bad kind / theta");
    DISPATCH(h, project_raw, h, x, n_rows, ld, kind, mean, stdv, out);
}

}  // extern "C"
