// lcx_outputs.hip - get_covariance, predict, invert (include/lcx.h): rank-n_hidden products whose output leaves the device
#include "engine.hpp"

extern "C" {

int lcx_covariance_rows_syn(lcx_ctx* h, const void* std_host, int64_t row0, int64_t nrows, void* out) {
    NEED(h);
    if (!std_host || !out || row0 < 0 || nrows < 1 || row0 + nrows > h->V) return fail(LCX_ERR_ARG, "lcx_covariance_rows_syn: bad range");
    DISPATCH(h, covariance_syn, h, std_host, row0, nrows, out);
}

int lcx_covariance_rows(lcx_ctx* h, double eps, const void* std_host, int64_t row0, int64_t nrows, void* out) {
    NEED(h);
    if (!std_host || !out || row0 < 0 || nrows < 1 || row0 + nrows > h->V) return fail(LCX_ERR_ARG, "lcx_covariance_rows: bad range");
    DISPATCH(h, covariance, h, eps, std_host, row0, nrows, out);
}

int lcx_covariance(lcx_ctx* h, int synergistic, double eps, const void* std_host, void* out, int64_t ld_out, double* kernel_seconds) {
    NEED(h);
    if (!std_host || !out || ld_out < h->V) return fail(LCX_ERR_ARG, "lcx_covariance: bad argument");
    DISPATCH(h, covariance_full, h, synergistic, eps, std_host, out, ld_out, kernel_seconds);
}

int lcx_predict(lcx_ctx* h, const void* y_host, int64_t n_rows, int synergistic, const void* xz_host, int kind, const void* mean,
                const void* stdv, void* out, int64_t ld_out, double* kernel_seconds) {
    NEED(h);
    if (!y_host || !out || n_rows < 1 || ld_out < h->V) return fail(LCX_ERR_ARG, "lcx_predict: bad argument");
    if (kind < 0 || kind > 2 || (kind != 0 && (!mean || !stdv))) return fail(LCX_ERR_ARG, "lcx_predict: bad kind / theta");
    DISPATCH(h, predict, h, y_host, n_rows, synergistic, xz_host, kind, mean, stdv, out, ld_out, kernel_seconds);
}

int lcx_invert(lcx_ctx* h, const void* x_host, int64_t n_rows, int64_t ld, int kind, const void* mean, const void* stdv, void* out,
               int64_t ld_out) {
    NEED(h);
    if (!x_host || !out || n_rows < 1 || ld < h->V || ld_out < h->V) return fail(LCX_ERR_ARG, "lcx_invert: bad argument");
    if (kind < 0 || kind > 2 || (kind != 0 && (!mean || !stdv))) return fail(LCX_ERR_ARG, "lcx_invert: bad kind / theta");
    DISPATCH(h, invert_rows, h, x_host, n_rows, ld, kind, mean, stdv, out, ld_out);
}

}  // extern "C"
