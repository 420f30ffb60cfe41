// levels_typed.hpp - the typed halves of lcx_levels.hip.  The moment / update levels instantiate the X-streaming kernels for seven
// factor paddings in two precisions: most of the library's compile time.  They are compiled as two translation units, one per
// precision (lcx_levels_f32.hip / lcx_levels_f64.hip = levels_typed.inc with LCX_T = float / double); lcx_levels.hip holds the entry
// points of the C ABI and picks the half by the handle's dtype.
#pragma once
#include "engine.hpp"

// name, parameters, arguments of every Impl<T, CT> level the entry points dispatch to
#define LCX_LEVEL_TABLE(X) \
    X(geometry, (lcx_ctx* h), (h)) \
    X(set_ws, (lcx_ctx* h, const void* w), (h, w)) \
    X(permute, (lcx_ctx* h, const int32_t* order), (h, order)) \
    X(moments_a, (lcx_ctx* h, int which), (h, which)) \
    X(moments_b, (lcx_ctx* h, int which, double eps, int quick), (h, which, eps, quick)) \
    X(moments_c, (lcx_ctx* h, int which), (h, which)) \
    X(update_a, (lcx_ctx* h), (h)) \
    X(update_b, (lcx_ctx* h, double eps), (h, eps)) \
    X(update_c, (lcx_ctx* h, double eps), (h, eps)) \
    X(make_trial, (lcx_ctx* h, double eta), (h, eta)) \
    X(trial_linear_a, (lcx_ctx* h, double eta), (h, eta)) \
    X(trial_linear_b, (lcx_ctx* h, double eps, double eta), (h, eps, eta)) \
    X(iterate, (lcx_ctx* h, double eps, double tol, double tc_cur, int more, double* out8), (h, eps, tol, tc_cur, more, out8)) \
    X(syn_moments_b, (lcx_ctx* h, int which, double yscale), (h, which, yscale)) \
    X(syn_moments_c, (lcx_ctx* h, int which), (h, which)) \
    X(syn_update_a, (lcx_ctx* h), (h)) \
    X(syn_update_b, (lcx_ctx* h, double eta), (h, eta)) \
    X(rescale, (lcx_ctx* h, double e0, double e1), (h, e0, e1)) \
    X(init_scale, (lcx_ctx* h), (h)) \
    X(get_moment, (lcx_ctx* h, int which, int key, double eps, void* out), (h, which, key, eps, out)) \
    X(set_moment, (lcx_ctx* h, int which, int key, const void* in), (h, which, key, in)) \
    X(split_supported, (lcx_ctx* h), (h)) \
    X(kernel_name, (lcx_ctx* h, int kind, char* buf, int64_t len), (h, kind, buf, len))

#define LCX_LEVEL_DECL(name, params, args) int name params;
#pragma GCC visibility push(hidden)          // internal to the library: the boundary is include/lcx.h
namespace lcx_f32 {
LCX_LEVEL_TABLE(LCX_LEVEL_DECL)
int get_ws(lcx_ctx* h, int which, void* w);        // Impl::fetch_mv of the weights (typed pointers)
int detail(lcx_ctx* h, int which);                 // Impl::detail into the set's own arrays
}
namespace lcx_f64 {
LCX_LEVEL_TABLE(LCX_LEVEL_DECL)
int get_ws(lcx_ctx* h, int which, void* w);
int detail(lcx_ctx* h, int which);
}
#pragma GCC visibility pop
#undef LCX_LEVEL_DECL
