// gemm_probe6.hip - float64 contraction on v_mfma_f64_4x4x4 (gemm_tn4_kernel) vs the 16x16x4 kernel (GPU box only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
#include "probe_kernels.hpp"
using namespace lcx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; int slots; };

template <int CT, int RT, int KW, int U, bool NT = false, bool SERIAL = false>
Variant mk4(const double* A, int64_t lda, int64_t K, int64_t vcols, const double* B, double* out, int S) {
    auto kern = gemm_tn4_kernel<CT, RT, KW, U, NT, SERIAL>;
    const size_t lds = Tn4Lds<CT, RT, KW, U, SERIAL>::bytes;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    char buf[200];
    snprintf(buf, 200, "tn4 (4x4x4) RT=%d KW=%d U=%d NT=%d SER=%d S=%d blocks=%d bpc=%d lds=%zu", RT, KW, U, (int)NT, (int)SERIAL, S, (int)(vcols / (16 * RT)) * S, bpc, lds);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
}
// round 4: the A stream prefetched two groups ahead (gemm_tn4r_kernel, probe_kernels.hpp)
template <int CT, int RT, int KW, int U, bool NT = false>
Variant mk4r(const double* A, int64_t lda, int64_t K, int64_t vcols, const double* B, double* out, int S) {
    auto kern = gemm_tn4r_kernel<CT, RT, KW, U, NT>;
    const size_t lds = Tn4Lds<CT, RT, KW, U, false>::bytes;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    char buf[200];
    snprintf(buf, 200, "tn4r (A ring of 3) RT=%d KW=%d U=%d NT=%d S=%d blocks=%d bpc=%d lds=%zu", RT, KW, U, (int)NT, S, (int)(vcols / (16 * RT)) * S, bpc, lds);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
}
template <int CT, int RT, int KW, int U, int SCHED>
Variant mk4s(const double* A, int64_t lda, int64_t K, int64_t vcols, const double* B, double* out, int S) {
    auto kern = gemm_tn4s_kernel<CT, RT, KW, U, true, SCHED>;
    const size_t lds = Tn4Lds<CT, RT, KW, U, false>::bytes;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    char buf[200];
    snprintf(buf, 200, "tn4s (interleaved schedule %d) RT=%d KW=%d U=%d S=%d blocks=%d bpc=%d", SCHED, RT, KW, U, S, (int)(vcols / (16 * RT)) * S, bpc);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
}
template <int CT, int RT, int KW>
Variant mkprod(const double* A, int64_t lda, int64_t K, int64_t vcols, const double* B, double* out, int S) {
    auto kern = gemm_tn_probe_kernel<double, CT, RT, KW, false, 0, 4>;
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * 8;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char buf[200];
    snprintf(buf, 200, "tn (16x16x4) RT=%d KW=%d S=%d", RT, KW, S);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, (int64_t)(16 * RT), B, (const double*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
}
static void bench(std::vector<Variant>& vs, double gbytes, double tflop, int rounds = 7, int iters = 10) {
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
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-64s med %7.1f us (min %7.1f max %7.1f) %6.0f GB/s %5.1f TF/s\n", v.name.c_str(), med * 1e3, v.ms.front() * 1e3, v.ms.back() * 1e3,
               gbytes / med * 1e3, tflop / med * 1e3);
    }
    fflush(stdout);
}
template <int CT, int RT>
void suite(const char* name, int64_t K, int64_t V, std::initializer_list<int> splits) {
    const int Mp = 16 * CT;
    double *A, *B, *out;
    CK(hipMalloc(&A, 8 * K * V)); CK(hipMalloc(&B, 8 * K * Mp)); CK(hipMalloc(&out, 8 * 40 * V * Mp));
    std::vector<double> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), 8 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 8 * K * Mp, hipMemcpyHostToDevice));
    const double gb = 8 * ((double)K * V + (double)Mp * (K + V)) / 1e9, tf = 2.0 * K * V * Mp / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d f64\n", name, (long)K, (long)V, Mp);
    std::vector<Variant> vs;
    vs.push_back(mkprod<CT, RT, 4>(A, V, K, V, B, out, *splits.begin()));
    for (int S : splits) {
        vs.push_back(mk4<CT, RT, 4, 4>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, RT, 4, 2>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, RT, 8, 2>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, RT, 2, 4>(A, V, K, V, B, out, S));
    }
    const size_t n1 = (size_t)V * Mp;
    vs[0].launch(); CK(hipDeviceSynchronize());
    std::vector<double> r((size_t)vs[0].slots * n1);
    CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
    for (size_t vi = 1; vi < vs.size(); vi += 1) {
        CK(hipMemset(out, 0xff, 8 * (size_t)vs[vi].slots * n1));
        vs[vi].launch(); CK(hipDeviceSynchronize());
        std::vector<double> o((size_t)vs[vi].slots * n1);
        CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s2 = 0; s2 < vs[vi].slots; ++s2) so += o[s2 * n1 + x];
            for (int s2 = 0; s2 < vs[0].slots; ++s2) sr += r[s2 * n1 + x];
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        if (vi < 5 || md > 1e-9 * mx) printf("check %-60s max |diff| %.3e (max |ref| %.3e) %s\n", vs[vi].name.c_str(), md, mx, md <= 1e-11 * mx ? "ok" : "FAIL");
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}
template <int CT>
void suite8(const char* name, int64_t K, int64_t V, std::initializer_list<int> splits) {
    const int Mp = 16 * CT;
    double *A, *B, *out;
    CK(hipMalloc(&A, 8 * K * V)); CK(hipMalloc(&B, 8 * K * Mp)); CK(hipMalloc(&out, 8 * 40 * V * Mp));
    std::vector<double> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), 8 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 8 * K * Mp, hipMemcpyHostToDevice));
    const double gb = 8 * ((double)K * V + (double)Mp * (K + V)) / 1e9, tf = 2.0 * K * V * Mp / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d f64 (wide tiles)\n", name, (long)K, (long)V, Mp);
    std::vector<Variant> vs;
    vs.push_back(mkprod<CT, 4, 4>(A, V, K, V, B, out, 6));
    for (int S : splits) {
        vs.push_back(mk4<CT, 4, 4, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 4, 4, 4, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 8, 4, 2, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 8, 4, 4, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 8, 8, 2, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 4, 8, 4, true, true>(A, V, K, V, B, out, S));
    }
    const size_t n1 = (size_t)V * Mp;
    vs[0].launch(); CK(hipDeviceSynchronize());
    std::vector<double> r((size_t)vs[0].slots * n1);
    CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
    for (size_t vi = 1; vi < 7; ++vi) {
        CK(hipMemset(out, 0xff, 8 * (size_t)vs[vi].slots * n1));
        vs[vi].launch(); CK(hipDeviceSynchronize());
        std::vector<double> o((size_t)vs[vi].slots * n1);
        CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s2 = 0; s2 < vs[vi].slots; ++s2) so += o[s2 * n1 + x];
            for (int s2 = 0; s2 < vs[0].slots; ++s2) sr += r[s2 * n1 + x];
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        printf("check %-66s max |diff| %.3e %s\n", vs[vi].name.c_str(), md, md <= 1e-11 * mx ? "ok" : "FAIL");
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}
// more resident waves per SIMD: shorter register double buffers (U = 1, 2) and the serial LDS reduction (one tile of LDS
// instead of KW) so that 3-4 blocks fit per CU
template <int CT>
void suite_occ(const char* name, int64_t K, int64_t V, std::initializer_list<int> splits) {
    const int Mp = 16 * CT;
    double *A, *B, *out;
    CK(hipMalloc(&A, 8 * K * V)); CK(hipMalloc(&B, 8 * K * Mp)); CK(hipMalloc(&out, 8 * 40 * V * Mp));
    std::vector<double> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), 8 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 8 * K * Mp, hipMemcpyHostToDevice));
    const double gb = 8 * ((double)K * V + (double)Mp * (K + V)) / 1e9, tf = 2.0 * K * V * Mp / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d f64 (occupancy variants)\n", name, (long)K, (long)V, Mp);
    std::vector<Variant> vs;
    vs.push_back(mk4<CT, 4, 4, 4, true>(A, V, K, V, B, out, *splits.begin()));
    for (int S : splits) {
        vs.push_back(mk4<CT, 4, 4, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 4, 4, 2, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 4, 4, 1, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 2, 4, 4, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 2, 4, 2, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<CT, 4, 2, 2, true, true>(A, V, K, V, B, out, S));
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

// round 4: config 2's two passes from ONE panel-major copy (gemm_tn4p / gemm_tn4c, probe_kernels.hpp) vs the two-copy production layout
template <int CT, int RT, int KW, int U, bool NT, bool CSTYLE>
Variant mk4panel(const double* XP, int64_t nrows_pad, int64_t K, int64_t out_rows, const double* B, double* out, int S) {
    const size_t lds = Tn4Lds<CT, RT, KW, U, false>::bytes;
    char buf[200];
    int bpc = 0;
    if constexpr (CSTYLE) {
        auto kern = gemm_tn4c_kernel<CT, RT, KW, U, NT>;
        if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
        snprintf(buf, 200, "tn4c X.B^T from the PANEL copy RT=%d KW=%d U=%d NT=%d S=%d blocks=%d bpc=%d", RT, KW, U, (int)NT, S, (int)(out_rows / (16 * RT)) * S, bpc);
        return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(out_rows / (16 * RT)), S), dim3(64 * KW), lds, 0, XP, nrows_pad * 8, B, out, out_rows, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
    } else {
        auto kern = gemm_tn4p_kernel<CT, RT, KW, U, NT>;
        if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
        snprintf(buf, 200, "tn4p X^T.Y from the PANEL copy RT=%d KW=%d U=%d NT=%d S=%d blocks=%d bpc=%d", RT, KW, U, (int)NT, S, (int)(out_rows / (16 * RT)) * S, bpc);
        return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(out_rows / (16 * RT)), S), dim3(64 * KW), lds, 0, XP, nrows_pad * 8, B, out, out_rows, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
    }
}
__global__ void transpose_f64_kernel(const double* X, int64_t ldx, double* XT, int64_t ldt, int64_t rows, int64_t cols) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < rows * cols; k += (int64_t)gridDim.x * blockDim.x)
        XT[(k % cols) * ldt + k / cols] = X[(k / cols) * ldx + k % cols];
}
static void compare(const char* what, std::vector<Variant>& vs, double* out, size_t n1) {
    vs[0].launch(); CK(hipDeviceSynchronize());
    std::vector<double> r((size_t)vs[0].slots * n1);
    CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
    for (size_t vi = 1; vi < vs.size(); ++vi) {
        CK(hipMemset(out, 0xff, 8 * (size_t)vs[vi].slots * n1));
        vs[vi].launch(); CK(hipDeviceSynchronize());
        std::vector<double> o((size_t)vs[vi].slots * n1);
        CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s2 = 0; s2 < vs[vi].slots; ++s2) so += o[s2 * n1 + x];
            for (int s2 = 0; s2 < vs[0].slots; ++s2) sr += r[s2 * n1 + x];
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        printf("check %s %-72s max |diff| %.3e %s\n", what, vs[vi].name.c_str(), md, md <= 1e-11 * mx ? "ok" : "FAIL");
    }
}
void suite_c2panel(int64_t N, int64_t V) {
    constexpr int CT = 2;
    const int Mp = 16 * CT;
    double *X, *XT, *XP, *B, *out;
    const int64_t big = std::max(N, V);
    CK(hipMalloc(&X, 8 * N * V)); CK(hipMalloc(&XT, 8 * N * V)); CK(hipMalloc(&XP, 8 * N * V));
    CK(hipMalloc(&B, 8 * big * Mp)); CK(hipMalloc(&out, 8 * 40 * big * Mp));
    std::vector<double> h((size_t)N * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(X, h.data(), 8 * N * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 8 * big * Mp, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(transpose_f64_kernel, dim3(4096), dim3(256), 0, 0, X, V, XT, N, N, V);
    hipLaunchKernelGGL((panelize_kernel<double>), dim3(4096), dim3(256), 0, 0, X, V, XP, N, V);
    CK(hipDeviceSynchronize());
    const double gb = 8 * ((double)N * V + (double)Mp * (N + V)) / 1e9, tf = 2.0 * N * V * Mp / 1e12;
    {
        printf("== c2 X^T.Y: X %ld x %ld f64, 32 factors: row-major X (production tn4) vs the panel copy (tn4p)\n", (long)N, (long)V);
        std::vector<Variant> vs;
        vs.push_back(mk4<CT, 4, 4, 4, true>(X, V, N, V, B, out, 3));
        vs.push_back(mk4panel<CT, 4, 4, 4, true, false>(XP, N, N, V, B, out, 3));
        vs.push_back(mk4panel<CT, 4, 4, 4, false, false>(XP, N, N, V, B, out, 3));
        vs.push_back(mk4<CT, 4, 4, 4, true>(X, V, N, V, B, out, 6));
        vs.push_back(mk4panel<CT, 4, 4, 4, true, false>(XP, N, N, V, B, out, 6));
        compare("xty", vs, out, (size_t)V * Mp);
        bench(vs, gb, tf);
    }
    {
        printf("== c2 X.B^T: transposed copy (production tn4) vs the panel copy (tn4c)\n");
        std::vector<Variant> vs;
        vs.push_back(mk4<CT, 4, 4, 4, true>(XT, N, V, N, B, out, 3));
        vs.push_back(mk4panel<CT, 4, 4, 4, true, true>(XP, N, V, N, B, out, 3));
        vs.push_back(mk4panel<CT, 4, 4, 4, false, true>(XP, N, V, N, B, out, 3));
        vs.push_back(mk4<CT, 4, 4, 4, true>(XT, N, V, N, B, out, 2));
        vs.push_back(mk4panel<CT, 4, 4, 4, true, true>(XP, N, V, N, B, out, 2));
        vs.push_back(mk4panel<CT, 4, 4, 2, true, true>(XP, N, V, N, B, out, 3));
        compare("xbt", vs, out, (size_t)N * Mp);
        bench(vs, gb, tf);
    }
    CK(hipFree(X)); CK(hipFree(XT)); CK(hipFree(XP)); CK(hipFree(B)); CK(hipFree(out));
}

template <int CT>
void suite_ring(const char* name, int64_t K, int64_t V, std::initializer_list<int> splits) {
    const int Mp = 16 * CT;
    double *A, *B, *out;
    CK(hipMalloc(&A, 8 * K * V)); CK(hipMalloc(&B, 8 * K * Mp)); CK(hipMalloc(&out, 8 * 40 * V * Mp));
    std::vector<double> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), 8 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 8 * K * Mp, hipMemcpyHostToDevice));
    const double gb = 8 * ((double)K * V + (double)Mp * (K + V)) / 1e9, tf = 2.0 * K * V * Mp / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d f64 (A prefetched two groups ahead)\n", name, (long)K, (long)V, Mp);
    std::vector<Variant> vs;
    for (int S : splits) {
        vs.push_back(mk4<CT, 4, 4, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4s<CT, 4, 4, 4, 1>(A, V, K, V, B, out, S));
        vs.push_back(mk4s<CT, 4, 4, 4, 2>(A, V, K, V, B, out, S));
        vs.push_back(mk4r<CT, 4, 4, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4r<CT, 4, 4, 2, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4r<CT, 4, 8, 2, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4r<CT, 2, 4, 4, true>(A, V, K, V, B, out, S));
    }
    const size_t n1 = (size_t)V * Mp;
    vs[0].launch(); CK(hipDeviceSynchronize());
    std::vector<double> r((size_t)vs[0].slots * n1);
    CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
    for (size_t vi = 1; vi < 7; ++vi) {
        CK(hipMemset(out, 0xff, 8 * (size_t)vs[vi].slots * n1));
        vs[vi].launch(); CK(hipDeviceSynchronize());
        std::vector<double> o((size_t)vs[vi].slots * n1);
        CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s2 = 0; s2 < vs[vi].slots; ++s2) so += o[s2 * n1 + x];
            for (int s2 = 0; s2 < vs[0].slots; ++s2) sr += r[s2 * n1 + x];
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        printf("check %-66s max |diff| %.3e %s\n", vs[vi].name.c_str(), md, md <= 1e-11 * mx ? "ok" : "FAIL");
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

// round 3: a merged pass for float64 with <= 32 padded factors - X.[grad | ws + update]^T as ONE 64-column contraction
// (two 32-column passes read X twice; at config 2 the pass is HBM-bound, so the second read is the cost).  Reference row:
// the production 32-column kernel; the 64-column candidates must beat TWICE its time.
void suite_merged(const char* name, int64_t K, int64_t V, std::initializer_list<int> splits) {
    double *A, *B, *out;
    CK(hipMalloc(&A, 8 * K * V)); CK(hipMalloc(&B, 8 * K * 64)); CK(hipMalloc(&out, 8 * 40 * V * 64));
    std::vector<double> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), 8 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 8 * K * 64, hipMemcpyHostToDevice));
    const double gb = 8 * ((double)K * V + 64.0 * (K + V)) / 1e9, tf = 2.0 * K * V * 64 / 1e12;
    printf("== %s: K=%ld V=%ld f64; first row = the 32-column production kernel (GB/s, TF/s columns are for 64 columns)\n", name, (long)K, (long)V);
    std::vector<Variant> vs;
    vs.push_back(mk4<2, 4, 4, 4, true>(A, V, K, V, B, out, *splits.begin()));
    for (int S : splits) {
        vs.push_back(mk4<4, 4, 4, 4, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<4, 4, 4, 2, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<4, 2, 4, 4, true, false>(A, V, K, V, B, out, S));
        vs.push_back(mk4<4, 2, 4, 4, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<4, 2, 4, 2, true, true>(A, V, K, V, B, out, S));
        vs.push_back(mk4<4, 2, 8, 2, true, true>(A, V, K, V, B, out, S));
    }
    // correctness of the 64-column variants against the 16x16x4 kernel
    std::vector<Variant> ref;
    ref.push_back(mkprod<4, 2, 4>(A, V, K, V, B, out, 3));
    const size_t n1 = (size_t)V * 64;
    ref[0].launch(); CK(hipDeviceSynchronize());
    std::vector<double> r((size_t)ref[0].slots * n1);
    CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
    for (size_t vi = 1; vi < 7; ++vi) {
        CK(hipMemset(out, 0xff, 8 * (size_t)vs[vi].slots * n1));
        vs[vi].launch(); CK(hipDeviceSynchronize());
        std::vector<double> o((size_t)vs[vi].slots * n1);
        CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s2 = 0; s2 < vs[vi].slots; ++s2) so += o[s2 * n1 + x];
            for (int s2 = 0; s2 < ref[0].slots; ++s2) sr += r[s2 * n1 + x];
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        printf("check %-66s max |diff| %.3e %s\n", vs[vi].name.c_str(), md, md <= 1e-11 * mx ? "ok" : "FAIL");
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "merged") {
        suite_merged("c2_x_gw (X.[grad | ws+update]^T, contraction over the 5056 variables)", 5056, 10048, {1, 2, 3, 4});
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "c2panel") {
        suite_c2panel(10048, 5056);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "ring") {
        suite_ring<2>("c2_xty", 10048, 5120, {3, 6});
        suite_ring<2>("c2_xw", 5120, 10112, {2, 3});
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "occ") {
        suite_occ<2>("c2_xty", 10048, 5120, {3, 4, 6, 9, 12});
        suite_occ<2>("c2_xw", 5120, 10112, {2, 3, 4, 6});
        return 0;
    }
    suite8<2>("c2_xty_128", 10048, 5120, {3, 6, 9, 12});
    suite8<2>("c2_xw_128", 5120, 10112, {2, 3, 6});
    return 0;
}
