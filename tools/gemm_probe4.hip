// gemm_probe4.hip - production gemm_ct_kernel (column-tiled waves, B through LDS, stream-K) vs gemm_tn_kernel,
// interleaved medians + correctness (GPU box only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
#include <string.h>
#include "probe_kernels.hpp"
using namespace lcx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one wave on one CU, on a second stream: effective shader clock (clock64 vs the 100 MHz wall clock) while the
// measured kernel owns the rest of the chip
__global__ void clock_sampler(long long* out, long long wall_ticks) {
    const long long r0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - r0 < wall_ticks) __builtin_amdgcn_s_sleep(32);
    out[0] = clock64() - c0;
    out[1] = wall_clock64() - r0;
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; int maxslots; };

template <typename T, int CT, int RT, int KW, int U, bool NT = false, int PRIO = 0>
Variant mkct(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_ct_probe_kernel<T, CT, RT, KW, U, NT, PRIO>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((vcols + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    char buf[200];
    snprintf(buf, 200, "ct RT=%d KW=%d U=%d NT=%d PRIO=%d bpc=%d(use %d) nb=%d nsuper=%d slots=%d", RT, KW, U, (int)NT, PRIO, bpc, use, nb, nsuper, maxslots);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, A, lda, B, out, vcols, vcols, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

// round 3: gemm_cr (A row-major, contraction along its contiguous axis: X.B^T from X itself, no transposed copy)
template <typename T, int CT, int RT, int KW, int U, bool NT = false>
Variant mkcr(const T* A, int64_t lda, int64_t K, int64_t nrows, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_cr_kernel<T, CT, RT, KW, U, NT>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((nrows + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    char buf[200];
    snprintf(buf, 200, "cr (row-major A) RT=%d KW=%d U=%d NT=%d bpc=%d(use %d) nb=%d nsuper=%d slots=%d", RT, KW, U, (int)NT, bpc, use, nb, nsuper, maxslots);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, A, lda, B, out, nrows, nrows, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

// round 4: gemm_cr with other A-load shapes (probe_kernels.hpp, gemm_cr2_kernel)
template <typename T, int CT, int RT, int KW, int U, int LOAD>
Variant mkcr2(const T* A, int64_t lda, int64_t K, int64_t nrows, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_cr2_kernel<T, CT, RT, KW, U, LOAD>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((nrows + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    static const char* what[] = {"production mapping", "quad loads + ds_bpermute", "quad loads, NO permute (timing only)", "quad loads + LDS strip"};
    char buf[200];
    snprintf(buf, 200, "cr2 %-36s U=%d bpc=%d(use %d) slots=%d", what[LOAD], U, bpc, use, maxslots);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, A, lda, B, out, nrows, nrows, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

template <typename T> __global__ void transpose_probe_kernel_t(const T* X, int64_t ldx, T* XT, int64_t ldt, int64_t rows, int64_t cols) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < rows * cols; k += (int64_t)gridDim.x * blockDim.x)
        XT[(k % cols) * ldt + k / cols] = X[(k / cols) * ldx + k % cols];
}

template <typename T, int CT, int RT, int KW>
Variant mkprod(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int S) {
    auto kern = gemm_tn_probe_kernel<T, CT, RT, KW, false, 0, 4>;
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char buf[200];
    snprintf(buf, 200, "tn RT=%d KW=%d S=%d blocks=%d", RT, KW, S, (int)(vcols / (16 * RT)) * S);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, (int64_t)(16 * RT), B, (const T*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
}

static void bench(std::vector<Variant>& vs, double gbytes, double tflop, int rounds = 5, int iters = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& v : vs) v.launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            v.launch();
            CK(hipEventRecord(a, 0));
            for (int it = 0; it < iters; ++it) v.launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            v.ms.push_back(ms / iters);
        }
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    long long* clk; CK(hipMalloc(&clk, 64));
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        // shader clock during 8 back-to-back launches
        CK(hipDeviceSynchronize());
        v.launch();
        hipLaunchKernelGGL(clock_sampler, dim3(1), dim3(64), 0, s2, clk, (long long)(med * 1e-3 * 6.0 * 1e8));
        for (int it = 0; it < 8; ++it) v.launch();
        CK(hipDeviceSynchronize());
        long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        printf("%-60s med %8.1f us (min %8.1f max %8.1f) %6.0f GB/s %6.1f TF/s  shader clk %4.0f MHz\n", v.name.c_str(), med * 1e3, v.ms.front() * 1e3,
               v.ms.back() * 1e3, gbytes / med * 1e3, tflop / med * 1e3, (double)hc[0] / (double)hc[1] * 100.0);
    }
    fflush(stdout);
}

template <typename T>
void check(const char* what, Variant& v, Variant& ref, T* out, T* refout, int64_t V, int Mp, double tol) {
    // ref writes into the same `out`; run ref first, copy, then v
    const size_t n1 = (size_t)V * Mp;
    ref.launch();
    CK(hipDeviceSynchronize());
    std::vector<T> r2((size_t)ref.maxslots * n1);
    CK(hipMemcpy(r2.data(), out, r2.size() * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemset(out, 0xff, sizeof(T) * v.maxslots * n1));
    v.launch();
    CK(hipDeviceSynchronize());
    std::vector<T> o((size_t)v.maxslots * n1);
    CK(hipMemcpy(o.data(), out, o.size() * sizeof(T), hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t x = 0; x < n1; ++x) {
        double so = 0, sr = 0;
        for (int s2 = 0; s2 < v.maxslots; ++s2) so += o[s2 * n1 + x];
        for (int s2 = 0; s2 < ref.maxslots; ++s2) sr += r2[s2 * n1 + x];
        md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
    }
    printf("check %-10s %-52s max |diff| = %.3e (max |ref| = %.3e) %s\n", what, v.name.c_str(), md, mx, md <= tol * mx ? "ok" : "FAIL");
}

// Y[N][Mp] = X[N][V] . B[V][Mp]: gemm_ct on XT ([V][N], contraction over its rows) vs gemm_cr on X ([N][V], contraction along rows)
template <typename T, int CT>
void suite_cr(const char* name, int64_t N, int64_t V) {
    T *X, *XT, *B, *out;
    CK(hipMalloc(&X, sizeof(T) * N * V));
    CK(hipMalloc(&XT, sizeof(T) * N * V));
    CK(hipMalloc(&B, sizeof(T) * V * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 40 * N * 16 * CT));
    {
        std::vector<T> h((size_t)4096 * 4096);
        for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
        for (size_t off = 0; off < (size_t)N * V; off += h.size())
            CK(hipMemcpy(X + off, h.data(), sizeof(T) * std::min(h.size(), (size_t)N * V - off), hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data() + 11, sizeof(T) * V * 16 * CT, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((transpose_probe_kernel_t<T>), dim3(4096), dim3(256), 0, 0, X, V, XT, N, N, V);
    CK(hipDeviceSynchronize());
    const double gb = sizeof(T) * ((double)N * V + 16.0 * CT * (N + V)) / 1e9, tf = 2.0 * N * V * 16 * CT / 1e12;
    printf("== %s: X %ld x %ld, Mp=%d elt=%zu: X.B^T through the transposed copy (ct) vs from X itself (cr)\n", name, (long)N, (long)V, 16 * CT, sizeof(T));
    constexpr int R = CtShape<T, CT>::RT;
    std::vector<Variant> vs;
    vs.push_back(mkct<T, CT, R, 4, 4, true>(XT, N, V, N, B, out, 2));
    vs.push_back(mkcr<T, CT, R, 4, 4, false>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr<T, CT, R, 4, 8, false>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr<T, CT, R, 4, 4, false>(X, V, V, N, B, out, 3));
    vs.push_back(mkcr<T, CT, R, 8, 4, false>(X, V, V, N, B, out, 1));
    if constexpr (CT <= 4 && sizeof(T) == 4) vs.push_back(mkcr<T, CT, R, 4, 16, false>(X, V, V, N, B, out, 2));
    if constexpr (CT <= 4) vs.push_back(mkcr<T, CT, 2 * R, 4, 4, false>(X, V, V, N, B, out, 2));
    const double tol = sizeof(T) == 8 ? 1e-12 : 2e-5;
    for (size_t k = 1; k < vs.size(); ++k) check<T>(name, vs[k], vs[0], out, out, N, 16 * CT, tol);
    bench(vs, gb, tf);
    CK(hipFree(X)); CK(hipFree(XT)); CK(hipFree(B)); CK(hipFree(out));
}

// round 4: both passes from ONE panel-major copy (the production kernels in PANEL mode: gemm_cr for X.B^T, gemm_ct for X^T.Y)
template <typename T, int CT, int RT, int KW, int U, bool NT>
Variant mkcrp(const T* XP, int64_t nrows_pad, int64_t K, int64_t nrows, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_cr_kernel<T, CT, RT, KW, U, NT, true>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((nrows + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    char buf[200];
    snprintf(buf, 200, "crp X.B^T from the PANEL-major copy RT=%d KW=%d U=%d NT=%d bpc=%d(use %d) slots=%d", RT, KW, U, (int)NT, bpc, use, maxslots);
    const int64_t ps = nrows_pad * PanelW<T>::v;
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, XP, ps, B, out, nrows, nrows, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}
template <typename T, int CT, int RT, int KW, int U, bool NT>
Variant mkctp(const T* XP, int64_t nrows_pad, int64_t K, int64_t vcols, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_ct_kernel<T, CT, RT, KW, U, NT, true>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((vcols + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    char buf[200];
    snprintf(buf, 200, "ctp X^T.Y from the PANEL-major copy RT=%d KW=%d U=%d NT=%d bpc=%d(use %d) slots=%d", RT, KW, U, (int)NT, bpc, use, maxslots);
    const int64_t ps = nrows_pad * PanelW<T>::v;
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, XP, ps, B, out, vcols, vcols, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

template <typename T, int CT>
void suite_panel(const char* name, int64_t N, int64_t V) {
    T *X, *XT, *XP, *B, *out;
    CK(hipMalloc(&X, sizeof(T) * N * V));
    CK(hipMalloc(&XT, sizeof(T) * N * V));
    CK(hipMalloc(&XP, sizeof(T) * N * V));
    const int64_t big = std::max(N, V);
    CK(hipMalloc(&B, sizeof(T) * big * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 40 * big * 16 * CT));
    {
        std::vector<T> h((size_t)4096 * 4096);
        for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
        for (size_t off = 0; off < (size_t)N * V; off += h.size())
            CK(hipMemcpy(X + off, h.data(), sizeof(T) * std::min(h.size(), (size_t)N * V - off), hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data() + 11, sizeof(T) * big * 16 * CT, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((transpose_probe_kernel_t<T>), dim3(4096), dim3(256), 0, 0, X, V, XT, N, N, V);
    hipLaunchKernelGGL((panelize_kernel<T>), dim3(4096), dim3(256), 0, 0, X, V, XP, N, V);
    CK(hipDeviceSynchronize());
    const double tol = sizeof(T) == 8 ? 1e-12 : 2e-5;
    constexpr int R = CtShape<T, CT>::RT;
    {
        const double gb = sizeof(T) * ((double)N * V + 16.0 * CT * (N + V)) / 1e9, tf = 2.0 * N * V * 16 * CT / 1e12;
        printf("== %s: X %ld x %ld, Mp=%d elt=%zu: X.B^T - transposed copy (ct) / row-major X (cr) / panel-major copy (crp)\n", name, (long)N, (long)V, 16 * CT, sizeof(T));
        std::vector<Variant> vs;
        vs.push_back(mkct<T, CT, R, 4, 4, true>(XT, N, V, N, B, out, 2));
        vs.push_back(mkcr<T, CT, R, 4, 4, false>(X, V, V, N, B, out, 2));
        vs.push_back(mkcrp<T, CT, R, 4, 4, false>(XP, N, V, N, B, out, 2));
        vs.push_back(mkcrp<T, CT, R, 4, 4, true>(XP, N, V, N, B, out, 2));
        vs.push_back(mkcrp<T, CT, R, 4, 8, true>(XP, N, V, N, B, out, 2));
        for (size_t k = 1; k < vs.size(); ++k) check<T>(name, vs[k], vs[0], out, out, N, 16 * CT, tol);
        bench(vs, gb, tf);
    }
    {
        const double gb = sizeof(T) * ((double)N * V + 16.0 * CT * (N + V)) / 1e9, tf = 2.0 * N * V * 16 * CT / 1e12;
        printf("== %s: X %ld x %ld, Mp=%d elt=%zu: X^T.Y - row-major X (ct) / panel-major copy (ctp)\n", name, (long)N, (long)V, 16 * CT, sizeof(T));
        std::vector<Variant> vs;
        vs.push_back(mkct<T, CT, R, 4, 4, true>(X, V, N, V, B, out, 2));
        vs.push_back(mkctp<T, CT, R, 4, 4, true>(XP, N, N, V, B, out, 2));
        vs.push_back(mkctp<T, CT, R, 4, 4, false>(XP, N, N, V, B, out, 2));
        for (size_t k = 1; k < vs.size(); ++k) check<T>(name, vs[k], vs[0], out, out, V, 16 * CT, tol);
        bench(vs, gb, tf);
    }
    CK(hipFree(X)); CK(hipFree(XT)); CK(hipFree(XP)); CK(hipFree(B)); CK(hipFree(out));
}

// geometry sweep of the two panel-layout passes (the balance moved: same speed as the two-copy layout at a 10 % higher shader clock)
template <typename T, int CT>
void suite_panel_sweep(const char* name, int64_t N, int64_t V) {
    T *X, *XP, *B, *out;
    CK(hipMalloc(&X, sizeof(T) * N * V));
    CK(hipMalloc(&XP, sizeof(T) * N * V));
    const int64_t big = std::max(N, V);
    CK(hipMalloc(&B, sizeof(T) * big * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 40 * big * 16 * CT));
    {
        std::vector<T> h((size_t)4096 * 4096);
        for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
        for (size_t off = 0; off < (size_t)N * V; off += h.size())
            CK(hipMemcpy(X + off, h.data(), sizeof(T) * std::min(h.size(), (size_t)N * V - off), hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data() + 11, sizeof(T) * big * 16 * CT, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((panelize_kernel<T>), dim3(4096), dim3(256), 0, 0, X, V, XP, N, V);
    CK(hipDeviceSynchronize());
    CK(hipFree(X));
    constexpr int R = CtShape<T, CT>::RT;
    const double gb = sizeof(T) * ((double)N * V + 16.0 * CT * (N + V)) / 1e9, tf = 2.0 * N * V * 16 * CT / 1e12;
    {
        printf("== %s sweep: X.B^T from the panel-major copy (crp)\n", name);
        std::vector<Variant> vs;
        vs.push_back(mkcrp<T, CT, R, 4, 4, true>(XP, N, V, N, B, out, 2));
        vs.push_back(mkcrp<T, CT, R, 4, 4, true>(XP, N, V, N, B, out, 3));
        vs.push_back(mkcrp<T, CT, R, 4, 4, true>(XP, N, V, N, B, out, 1));
        vs.push_back(mkcrp<T, CT, R, 4, 8, true>(XP, N, V, N, B, out, 2));
        vs.push_back(mkcrp<T, CT, R, 8, 4, true>(XP, N, V, N, B, out, 1));
        vs.push_back(mkcrp<T, CT, R, 2, 4, true>(XP, N, V, N, B, out, 4));
        vs.push_back(mkcrp<T, CT, R, 8, 8, true>(XP, N, V, N, B, out, 1));
        vs.push_back(mkcrp<T, CT, R, 4, 8, true>(XP, N, V, N, B, out, 3));
        if constexpr (CT <= 4) vs.push_back(mkcrp<T, CT, 2 * R, 4, 4, true>(XP, N, V, N, B, out, 2));
        if constexpr (CT <= 4) vs.push_back(mkcrp<T, CT, 2 * R, 4, 4, true>(XP, N, V, N, B, out, 1));
        bench(vs, gb, tf);
    }
    {
        printf("== %s sweep: X^T.Y from the panel-major copy (ctp)\n", name);
        std::vector<Variant> vs;
        vs.push_back(mkctp<T, CT, R, 4, 4, true>(XP, N, N, V, B, out, 2));
        vs.push_back(mkctp<T, CT, R, 4, 4, true>(XP, N, N, V, B, out, 3));
        vs.push_back(mkctp<T, CT, R, 4, 4, true>(XP, N, N, V, B, out, 1));
        vs.push_back(mkctp<T, CT, R, 4, 8, true>(XP, N, N, V, B, out, 2));
        vs.push_back(mkctp<T, CT, R, 8, 4, true>(XP, N, N, V, B, out, 1));
        vs.push_back(mkctp<T, CT, R, 2, 4, true>(XP, N, N, V, B, out, 4));
        vs.push_back(mkctp<T, CT, R, 8, 8, true>(XP, N, N, V, B, out, 1));
        vs.push_back(mkctp<T, CT, R, 4, 8, true>(XP, N, N, V, B, out, 3));
        bench(vs, gb, tf);
    }
    CK(hipFree(XP)); CK(hipFree(B)); CK(hipFree(out));
}

template <typename T, int CT>
void suite_cr2(const char* name, int64_t N, int64_t V) {
    T *X, *XT, *B, *out;
    CK(hipMalloc(&X, sizeof(T) * N * V));
    CK(hipMalloc(&XT, sizeof(T) * N * V));
    CK(hipMalloc(&B, sizeof(T) * V * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 40 * N * 16 * CT));
    {
        std::vector<T> h((size_t)4096 * 4096);
        for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
        for (size_t off = 0; off < (size_t)N * V; off += h.size())
            CK(hipMemcpy(X + off, h.data(), sizeof(T) * std::min(h.size(), (size_t)N * V - off), hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data() + 11, sizeof(T) * V * 16 * CT, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((transpose_probe_kernel_t<T>), dim3(4096), dim3(256), 0, 0, X, V, XT, N, N, V);
    CK(hipDeviceSynchronize());
    const double gb = sizeof(T) * ((double)N * V + 16.0 * CT * (N + V)) / 1e9, tf = 2.0 * N * V * 16 * CT / 1e12;
    printf("== %s: X %ld x %ld, Mp=%d elt=%zu: X.B^T from the row-major X with other A-load shapes (cr2) vs the transposed copy (ct)\n", name, (long)N,
           (long)V, 16 * CT, sizeof(T));
    constexpr int R = CtShape<T, CT>::RT;
    std::vector<Variant> vs;
    vs.push_back(mkct<T, CT, R, 4, 4, true>(XT, N, V, N, B, out, 2));
    vs.push_back(mkcr<T, CT, R, 4, 4, false>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr2<T, CT, R, 4, 4, 0>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr2<T, CT, R, 4, 4, 1>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr2<T, CT, R, 4, 4, 3>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr2<T, CT, R, 4, 8, 1>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr2<T, CT, R, 4, 8, 3>(X, V, V, N, B, out, 2));
    const size_t checked = vs.size();
    vs.push_back(mkcr2<T, CT, R, 4, 4, 2>(X, V, V, N, B, out, 2));
    vs.push_back(mkcr2<T, CT, R, 4, 8, 2>(X, V, V, N, B, out, 2));
    const double tol = sizeof(T) == 8 ? 1e-12 : 2e-5;
    for (size_t k = 1; k < checked; ++k) check<T>(name, vs[k], vs[0], out, out, N, 16 * CT, tol);
    bench(vs, gb, tf);
    CK(hipFree(X)); CK(hipFree(XT)); CK(hipFree(B)); CK(hipFree(out));
}

template <typename T, int CT, int TNRT>
void suite(const char* name, int64_t K, int64_t V, int tnS) {
    T *A, *B, *out;
    CK(hipMalloc(&A, sizeof(T) * K * V));
    CK(hipMalloc(&B, sizeof(T) * K * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 140 * V * 16 * CT));
    {
        std::vector<T> h((size_t)K * V);
        for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
        CK(hipMemcpy(A, h.data(), sizeof(T) * K * V, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data(), sizeof(T) * K * 16 * CT, hipMemcpyHostToDevice));
    }
    const double gb = sizeof(T) * ((double)K * V + 16.0 * CT * (K + V)) / 1e9, tf = 2.0 * K * V * 16 * CT / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d elt=%zu\n", name, (long)K, (long)V, 16 * CT, sizeof(T));
    std::vector<Variant> vs;
    vs.push_back(mkprod<T, CT, TNRT, 4>(A, V, K, V, B, out, tnS));
    constexpr int R = CtShape<T, CT>::RT;
    vs.push_back(mkct<T, CT, R, 4, 4, true>(A, V, K, V, B, out, 2));
    vs.push_back(mkct<T, CT, R, 4, 4, true, 1>(A, V, K, V, B, out, 2));
    vs.push_back(mkct<T, CT, R, 4, 4, true, 2>(A, V, K, V, B, out, 2));
    vs.push_back(mkct<T, CT, R, 4, 4, true, 1>(A, V, K, V, B, out, 3));
    const double tol = sizeof(T) == 8 ? 1e-12 : 2e-5;
    for (size_t k = 1; k < vs.size(); ++k) check<T>(name, vs[k], vs[0], out, out, V, 16 * CT, tol);
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(which, "all");
    if (!strcmp(which, "cr")) {
        suite_cr<float, 4>("c3", 50048, 100032);
        suite_cr<float, 8>("c4shard", 50048, 125056);
        suite_cr<double, 4>("c3f64", 50048, 50048);
        suite_cr<float, 2>("mid32f32", 20032, 20032);
        return 0;
    }
    if (!strcmp(which, "panelsweep")) {
        suite_panel_sweep<float, 4>("c3", 50048, 100032);
        suite_panel_sweep<float, 8>("c4shard", 50048, 125056);
        return 0;
    }
    if (!strcmp(which, "panel")) {
        suite_panel<float, 4>("c3", 50048, 100032);
        suite_panel<float, 8>("c4shard", 50048, 125056);
        suite_panel<double, 4>("c3f64", 50048, 50048);
        return 0;
    }
    if (!strcmp(which, "cr2")) {
        suite_cr2<float, 4>("c3", 50048, 100032);
        suite_cr2<float, 8>("c4shard", 50048, 125056);
        suite_cr2<double, 4>("c3f64", 50048, 50048);
        return 0;
    }
    if (all || !strcmp(which, "c3")) {
        suite<float, 4, 4>("c3l_xty", 50048, 20032, 3);
        suite<float, 4, 4>("c3l_xw", 20032, 50048, 2);
    }
    if (all || !strcmp(which, "c4")) {
        suite<float, 8, 2>("c4l_xty", 50048, 20032, 2);
        suite<float, 8, 2>("c4l_xw", 20032, 50048, 2);
    }
    if (all || !strcmp(which, "c2")) {
        suite<double, 2, 4>("c2_xty", 10048, 5056, 6);
        suite<double, 2, 4>("c2_xw", 5056, 10048, 3);
    }
    if (all || !strcmp(which, "odd")) {
        suite<float, 1, 4>("odd_f32_m16", 2048, 1984, 4);      // ragged super tile (1984 = 31 x 64)
        suite<double, 4, 2>("odd_f64_m64", 4096, 3008, 4);
        suite<double, 8, 1>("odd_f64_m128", 4096, 3008, 4);
        suite<float, 2, 4>("odd_f32_m32", 1024, 6464, 4);
    }
    return 0;
}
