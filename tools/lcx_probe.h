/*
 * lcx_probe.h - entry points of tools/liblcx_probe.so (the lab), NOT part of the drop-in boundary (include/lcx.h).
 *
 * liblcx_probe.so is the engine compiled a second time together with the hooks below (tools/lcx_probe.hip): kernel unit
 * tests and micro-benchmarks that need the engine's internals.  It exports everything include/lcx.h declares as well, so a
 * probe creates its handles through this library.
 */
#ifndef LCX_PROBE_H
#define LCX_PROBE_H

#include "../include/lcx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* micro-benchmark: `iters` back-to-back launches of one X-streaming GEMM (kind as above, with its
 * partial-sum reduction) on the resident X; returns the average wall time per launch from HIP events */
int lcx_bench_gemm(lcx_ctx* h, int kind, int iters, double* avg_ms);
/* experiment: the launches of one moment evaluation (lcx_moments_a + lcx_moments_b of set 1, one GPU) issued directly vs
 * captured into a hipGraph and replayed; average wall time per evaluation of each */
int lcx_bench_graph(lcx_ctx* h, double eps, int iters, double* direct_ms, double* graph_ms);

/* ---- kernel unit tests (parity of the two GEMM kernels in isolation) -------------------------- */
/* out (n_rows x m_pad) = A (n_rows x k, ld=lda) . B^T with B given as (k x m_pad) row-major     */
int lcx_test_gemm_nt(int dtype, int device, const void* a_host, int64_t n_rows, int64_t k, int64_t lda,
                     const void* b_host, int m_pad, void* out_host, int force_split, int force_kw);
/* out (v x m_pad) = A^T . B, A (k x v, ld=lda), B (k x m_pad)                                    */
int lcx_test_gemm_tn(int dtype, int device, const void* a_host, int64_t k, int64_t v, int64_t lda,
                     const void* b_host, int m_pad, const void* rowscale_host, void* out_host,
                     int force_split, int force_kw);

#ifdef __cplusplus
}
#endif
#endif /* LCX_PROBE_H */
