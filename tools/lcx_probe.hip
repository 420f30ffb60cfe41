// lcx_probe.hip - the lab: entry points that only tests and probes use, built into tools/liblcx_probe.so.
//
// The product library (linearcorex_amd/liblcx_hip.so, include/lcx.h) is the boundary a maintainer of the reference binds;
// it carries no micro-benchmarks and no kernel unit-test hooks.  tools/liblcx_probe.so = the engine's own objects (the same
// lcx_core / lcx_levels / lcx_data / lcx_outputs / empirical objects the product is linked from) + this unit's hooks, declared in
// tools/lcx_probe.h:
//   lcx_test_gemm_nt / lcx_test_gemm_tn   the X-streaming kernels in isolation (tests/test_gemm_kernels_gpu.py)
//   lcx_bench_gemm                        back-to-back launches of one pass on a handle's resident X (tools/gemm_sweep.py)
//   lcx_bench_graph                       one moment evaluation direct vs captured into a hipGraph (tools/graph_probe.py)
// A handle created through this library is a handle of this library: do not mix the two .so files on one handle.
#include "../linearcorex_amd/csrc/engine.hpp"
#include "lcx_probe.h"

// ---- isolated GEMM checks -------------------------------------------------------------------------
template <typename T, int CT>
static int test_nt(const void* a_host, int64_t n_rows, int64_t k, int64_t lda, const void* b_host, void* out_host,
                   int force_split, int force_kw) {
    constexpr int Mp = 16 * CT;
    const int64_t rows_pad = round_up(n_rows, 64), ldx = round_up(k, 64);
    T *xd, *bd, *od, *pd;
    hipStream_t st = 0;
    HIPCHECK(hipMalloc((void**)&xd, sizeof(T) * rows_pad * ldx));
    HIPCHECK(hipMalloc((void**)&bd, sizeof(T) * ldx * Mp));
    const int S = force_split > 0 ? force_split : 1, KW = force_kw > 0 ? force_kw : 4;
    HIPCHECK(hipMalloc((void**)&pd, sizeof(T) * S * rows_pad * Mp));
    HIPCHECK(hipMalloc((void**)&od, sizeof(T) * rows_pad * Mp));
    HIPCHECK(hipMemset(xd, 0, sizeof(T) * rows_pad * ldx));
    HIPCHECK(hipMemset(bd, 0, sizeof(T) * ldx * Mp));
    HIPCHECK(hipMemcpy2D(xd, ldx * sizeof(T), a_host, lda * sizeof(T), k * sizeof(T), n_rows, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bd, b_host, sizeof(T) * k * Mp, hipMemcpyHostToDevice));
    LCXCHECK((launch_nt<T, CT>(st, xd, ldx, rows_pad, bd, pd, S, KW, nullptr)));
    const int64_t n = rows_pad * Mp;
    hipLaunchKernelGGL((reduce_partials_kernel<T, T>), dim3(256), dim3(256), 0, st, pd, S, n, n, od, (const int*)nullptr);
    KCHECK();
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out_host, od, sizeof(T) * n_rows * Mp, hipMemcpyDeviceToHost));
    HIPCHECK(hipFree(xd)); HIPCHECK(hipFree(bd)); HIPCHECK(hipFree(od)); HIPCHECK(hipFree(pd));
    return LCX_OK;
}

template <typename T, int CT>
static int test_tn(const void* a_host, int64_t k, int64_t v, int64_t lda, const void* b_host, const void* rs_host,
                   void* out_host, int force_split, int force_kw) {
    constexpr int Mp = 16 * CT;
    const int64_t kpad = round_up(k, 64), ldv = round_up(v, 64);
    T *ad, *bd, *od, *pd, *sd = nullptr;
    hipStream_t st = 0;
    HIPCHECK(hipMalloc((void**)&ad, sizeof(T) * kpad * ldv));
    HIPCHECK(hipMalloc((void**)&bd, sizeof(T) * kpad * Mp));
    HIPCHECK(hipMalloc((void**)&sd, sizeof(T) * kpad));
    int S = force_split > 0 ? force_split : 1, KW = force_kw > 0 ? force_kw : 4;
    int ct_nb = 0, ct_ns = 0;
    if (force_kw == -1) {        // column-tiled stream-K kernel; force_split = number of blocks (0: as in production)
        if (rs_host) return fail(LCX_ERR_ARG, "gemm_ct has no row scale");
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
        ct_geometry<T, CT>(ncu, kpad, ldv, force_split, &ct_nb, &ct_ns, &S);
    }
    HIPCHECK(hipMalloc((void**)&pd, sizeof(T) * S * ldv * Mp));
    HIPCHECK(hipMalloc((void**)&od, sizeof(T) * ldv * Mp));
    HIPCHECK(hipMemset(pd, 0xff, sizeof(T) * S * ldv * Mp));     // every slot must be written by the kernel
    HIPCHECK(hipMemset(ad, 0, sizeof(T) * kpad * ldv));
    HIPCHECK(hipMemset(bd, 0, sizeof(T) * kpad * Mp));
    HIPCHECK(hipMemset(sd, 0, sizeof(T) * kpad));
    HIPCHECK(hipMemcpy2D(ad, ldv * sizeof(T), a_host, lda * sizeof(T), v * sizeof(T), k, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bd, b_host, sizeof(T) * k * Mp, hipMemcpyHostToDevice));
    if (force_kw == -2) {
        if constexpr (sizeof(T) == 8 && CT <= 2) {
            LCXCHECK((launch_tn4<CT>(st, (const double*)ad, ldv, kpad, ldv, (const double*)bd, (double*)pd, S, 4, nullptr)));
        } else {
            return fail(LCX_ERR_ARG, "gemm_tn4 is float64 with m_pad <= 32 only");
        }
    } else if (force_kw < 0) {
        LCXCHECK((launch_ct<T, CT>(st, ad, ldv, kpad, ldv, bd, pd, ct_nb, ct_ns, S, nullptr)));
    } else if (rs_host) {
        HIPCHECK(hipMemcpy(sd, rs_host, sizeof(T) * k, hipMemcpyHostToDevice));
        LCXCHECK((launch_tn<T, CT, TnShape<T, CT>::RT, true>(st, ad, ldv, kpad, ldv, bd, sd, pd, S, KW, nullptr)));
    } else {
        LCXCHECK((launch_tn<T, CT, TnShape<T, CT>::RT, false>(st, ad, ldv, kpad, ldv, bd, nullptr, pd, S, KW, nullptr)));
    }
    const int64_t n = ldv * Mp;
    hipLaunchKernelGGL((reduce_partials_kernel<T, T>), dim3(256), dim3(256), 0, st, pd, S, n, n, od, (const int*)nullptr);
    KCHECK();
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out_host, od, sizeof(T) * v * Mp, hipMemcpyDeviceToHost));
    HIPCHECK(hipFree(ad)); HIPCHECK(hipFree(bd)); HIPCHECK(hipFree(od)); HIPCHECK(hipFree(pd)); HIPCHECK(hipFree(sd));
    return LCX_OK;
}



extern "C" {

int lcx_bench_gemm(lcx_ctx* h, int kind, int iters, double* avg_ms) {
    NEED_MUT(h);
    if (kind < 0 || kind > 1 || iters < 1 || !avg_ms) return fail(LCX_ERR_ARG, "lcx_bench_gemm: bad argument");
    hipEvent_t a, b;
    HIPCHECK(hipEventCreate(&a));
    HIPCHECK(hipEventCreate(&b));
    auto once = [&]() -> int {
        if (kind == 0) { DISPATCH(h, nt_big, h, (const void*)h->Wt[0], nullptr, false); }
        DISPATCH(h, tn_big, h, nullptr);
    };
    for (int i = 0; i < 3; ++i) LCXCHECK(once());
    HIPCHECK(hipEventRecord(a, h->stream));
    for (int i = 0; i < iters; ++i) LCXCHECK(once());
    HIPCHECK(hipEventRecord(b, h->stream));
    HIPCHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, a, b));
    *avg_ms = ms / iters;
    HIPCHECK(hipEventDestroy(a));
    HIPCHECK(hipEventDestroy(b));
    return LCX_OK;
}

// Experiment: one moment evaluation of set 1 (the launches of lcx_moments_a + lcx_moments_b, one GPU) issued directly
// `iters` times vs captured once into a hipGraph and replayed `iters` times.  The replay publishes a stale sequence number,
// which is fine for timing; nothing reads the state in between.
int lcx_bench_graph(lcx_ctx* h, double eps, int iters, double* direct_ms, double* graph_ms) {
    NEED_MUT(h);
    if (iters < 1 || !direct_ms || !graph_ms) return fail(LCX_ERR_ARG, "lcx_bench_graph: bad argument");
    if (h->exchange) return fail(LCX_ERR_STATE, "lcx_bench_graph: one GPU only");
    hipEvent_t a, b;
    HIPCHECK(hipEventCreate(&a));
    HIPCHECK(hipEventCreate(&b));
    const bool timing = h->timing;
    h->timing = false;
    auto once = [&]() -> int {
        int rc = lcx_moments_a(h, 1);
        if (rc != LCX_OK) return rc;
        return lcx_moments_b(h, 1, eps, 1);
    };
    for (int i = 0; i < 3; ++i) LCXCHECK(once());
    HIPCHECK(hipEventRecord(a, h->stream));
    for (int i = 0; i < iters; ++i) LCXCHECK(once());
    HIPCHECK(hipEventRecord(b, h->stream));
    HIPCHECK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, a, b));
    *direct_ms = ms / iters;
    hipGraph_t graph;
    hipGraphExec_t exec;
    HIPCHECK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = once();
    hipError_t ce = hipStreamEndCapture(h->stream, &graph);
    if (rc != LCX_OK) return rc;
    HIPCHECK(ce);
    HIPCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) HIPCHECK(hipGraphLaunch(exec, h->stream));
    HIPCHECK(hipEventRecord(a, h->stream));
    for (int i = 0; i < iters; ++i) HIPCHECK(hipGraphLaunch(exec, h->stream));
    HIPCHECK(hipEventRecord(b, h->stream));
    HIPCHECK(hipEventSynchronize(b));
    HIPCHECK(hipEventElapsedTime(&ms, a, b));
    *graph_ms = ms / iters;
    HIPCHECK(hipGraphExecDestroy(exec));
    HIPCHECK(hipGraphDestroy(graph));
    HIPCHECK(hipEventDestroy(a));
    HIPCHECK(hipEventDestroy(b));
    h->timing = timing;
    // leave a consistent publication state behind: one more direct evaluation
    LCXCHECK(once());
    HIPCHECK(hipStreamSynchronize(h->stream));
    return LCX_OK;
}

int lcx_test_gemm_nt(int dtype, int device, const void* a, int64_t n_rows, int64_t k, int64_t lda, const void* b,
                     int m_pad, void* out, int fs, int fk) {
    HIPCHECK(hipSetDevice(device));
    const int ct = m_pad / 16;
    if (dtype == LCX_F32) {
        switch (ct) { case 1: return test_nt<float,1>(a,n_rows,k,lda,b,out,fs,fk); case 2: return test_nt<float,2>(a,n_rows,k,lda,b,out,fs,fk);
                      case 4: return test_nt<float,4>(a,n_rows,k,lda,b,out,fs,fk); case 8: return test_nt<float,8>(a,n_rows,k,lda,b,out,fs,fk); }
    } else {
        switch (ct) { case 1: return test_nt<double,1>(a,n_rows,k,lda,b,out,fs,fk); case 2: return test_nt<double,2>(a,n_rows,k,lda,b,out,fs,fk);
                      case 4: return test_nt<double,4>(a,n_rows,k,lda,b,out,fs,fk); case 8: return test_nt<double,8>(a,n_rows,k,lda,b,out,fs,fk); }
    }
    return fail(LCX_ERR_ARG, "m_pad must be 16, 32, 64 or 128 (the row-streaming kernel has no 256-factor form)");
}

int lcx_test_gemm_tn(int dtype, int device, const void* a, int64_t k, int64_t v, int64_t lda, const void* b, int m_pad,
                     const void* rs, void* out, int fs, int fk) {
    HIPCHECK(hipSetDevice(device));
    const int ct = m_pad / 16;
    if (dtype == LCX_F32) {
        switch (ct) { case 1: return test_tn<float,1>(a,k,v,lda,b,rs,out,fs,fk); case 2: return test_tn<float,2>(a,k,v,lda,b,rs,out,fs,fk);
                      case 4: return test_tn<float,4>(a,k,v,lda,b,rs,out,fs,fk); case 8: return test_tn<float,8>(a,k,v,lda,b,rs,out,fs,fk);
                      case 16: return test_tn<float,16>(a,k,v,lda,b,rs,out,fs,fk); }
    } else {
        switch (ct) { case 1: return test_tn<double,1>(a,k,v,lda,b,rs,out,fs,fk); case 2: return test_tn<double,2>(a,k,v,lda,b,rs,out,fs,fk);
                      case 4: return test_tn<double,4>(a,k,v,lda,b,rs,out,fs,fk); case 8: return test_tn<double,8>(a,k,v,lda,b,rs,out,fs,fk);
                      case 16: return test_tn<double,16>(a,k,v,lda,b,rs,out,fs,fk); }
    }
    return fail(LCX_ERR_ARG, "m_pad must be 16, 32, 64 or 128");
}


}  // extern "C"
