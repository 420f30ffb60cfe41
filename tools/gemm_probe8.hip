// gemm_probe8.hip - round 2: the float32 large-shard pass (BASELINE config 3: n_hidden 64) on other tile shapes.
//   ct   RT=4  production gemm_ct (64-column wave tiles, 256-column super tiles, v_mfma_f32_16x16x4)
//   ct   RT=8  128-column wave tiles / 512-column super tiles: half the B re-read and LDS reads per flop
//   ct32 NTB=n gemm_ct32 on v_mfma_f32_32x32x2 (same FLOP/clk, half the operand-register reads per flop), 32*n-column waves
// interleaved medians, shader clock under load, correctness against the production kernel (GPU box only).
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

__global__ void clock_sampler(long long* out, long long wall_ticks) {
    const long long r0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - r0 < wall_ticks) __builtin_amdgcn_s_sleep(32);
    out[0] = clock64() - c0;
    out[1] = wall_clock64() - r0;
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; int maxslots; };

template <typename T, int CT, int RT, int KW, int U, bool NT = true, int PRIO = 0>
Variant mkct(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_ct_probe_kernel<T, CT, RT, KW, U, NT, PRIO>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 && bpc_use < bpc ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((vcols + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)kern));
    char buf[240];
    snprintf(buf, 240, "ct   RT=%d KW=%d U=%d hint=%d bpc=%d(use %d) nb=%d nsuper=%d slots=%d regs=%d", RT, KW, U, PRIO, bpc, use, nb, nsuper, maxslots, fa.numRegs);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, A, lda, B, out, vcols, vcols, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

template <typename T, int CT, int RT, int KW, int U>
Variant mkct3(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int bpc_use = 0) {
    auto kern = gemm_ct3_kernel<T, CT, RT, KW, U, true>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 && bpc_use < bpc ? bpc_use : bpc;
    const int ng = (int)(K / (4 * U));
    const int nsuper = (int)((vcols + KW * 16 * RT - 1) / (KW * 16 * RT));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)kern));
    char buf[240];
    snprintf(buf, 240, "ct3  RT=%d KW=%d U=%d (A two groups ahead) bpc=%d(use %d) nb=%d nsuper=%d slots=%d regs=%d", RT, KW, U, bpc, use, nb, nsuper, maxslots, fa.numRegs);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, A, lda, B, out, vcols, vcols, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

template <int CT, int NTB, int KW, int U2, bool NT = true>
Variant mkct32(const float* A, int64_t lda, int64_t K, int64_t vcols, const float* B, float* out, int bpc_use = 0) {
    auto kern = gemm_ct32_kernel<CT, NTB, KW, U2, NT>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    const int use = bpc_use > 0 && bpc_use < bpc ? bpc_use : bpc;
    const int ng = (int)(K / (2 * U2));
    const int nsuper = (int)((vcols + KW * 32 * NTB - 1) / (KW * 32 * NTB));
    int64_t total = (int64_t)nsuper * ng;
    int nb = 256 * use;
    if (nb > total) nb = (int)total;
    const int maxslots = (nb + nsuper - 1) / nsuper + 1;
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)kern));
    char buf[240];
    snprintf(buf, 240, "ct32 NTB=%d KW=%d U2=%d bpc=%d(use %d) nb=%d nsuper=%d slots=%d regs=%d", NTB, KW, U2, bpc, use, nb, nsuper, maxslots, fa.numRegs);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, A, lda, B, out, vcols, vcols, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
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
        CK(hipDeviceSynchronize());
        v.launch();
        hipLaunchKernelGGL(clock_sampler, dim3(1), dim3(64), 0, s2, clk, (long long)(med * 1e-3 * 6.0 * 1e8));
        for (int it = 0; it < 8; ++it) v.launch();
        CK(hipDeviceSynchronize());
        long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        printf("%-78s med %8.1f us (min %8.1f max %8.1f) %6.0f GB/s %6.1f TF/s  shader clk %4.0f MHz\n", v.name.c_str(), med * 1e3, v.ms.front() * 1e3,
               v.ms.back() * 1e3, gbytes / med * 1e3, tflop / med * 1e3, (double)hc[0] / (double)hc[1] * 100.0);
    }
    fflush(stdout);
}

template <typename T>
void check(const char* what, Variant& v, Variant& ref, T* out, int64_t V, int Mp, double tol) {
    const size_t n1 = (size_t)V * Mp;
    CK(hipMemset(out, 0xff, sizeof(T) * ref.maxslots * n1));
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
    printf("check %-8s %-70s max |diff| = %.3e (max |ref| = %.3e) %s\n", what, v.name.c_str(), md, mx, (md <= tol * mx) ? "ok" : "FAIL");
}

template <int CT>
void suite(const char* name, int64_t K, int64_t V, bool small, bool zero = false) {
    typedef float T;
    T *A, *B, *out;
    CK(hipMalloc(&A, sizeof(T) * K * V));
    CK(hipMalloc(&B, sizeof(T) * K * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * (small ? 600 : 48) * V * 16 * CT));
    {
        std::vector<T> h((size_t)K * V);
        for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
        // zero: the same instructions on all-zero X - what the pass does when the data paths do not toggle (power headroom)
        if (zero) CK(hipMemset(A, 0, sizeof(T) * K * V)); else
        CK(hipMemcpy(A, h.data(), sizeof(T) * K * V, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data(), sizeof(T) * K * 16 * CT, hipMemcpyHostToDevice));
    }
    const double gb = sizeof(T) * ((double)K * V + 16.0 * CT * (K + V)) / 1e9, tf = 2.0 * K * V * 16 * CT / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d float32\n", name, (long)K, (long)V, 16 * CT);
    std::vector<Variant> vs;
    vs.push_back(mkct<T, CT, 4, 4, 4>(A, V, K, V, B, out, 2));          // production
    if constexpr (CT == 4) {
        vs.push_back(mkct<T, CT, 4, 4, 8>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, CT, 4, 4, 8>(A, V, K, V, B, out, 3));
        vs.push_back(mkct<T, CT, 8, 4, 4>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, CT, 8, 4, 4>(A, V, K, V, B, out, 1));
        vs.push_back(mkct<T, CT, 8, 2, 4>(A, V, K, V, B, out, 4));
        vs.push_back(mkct<T, CT, 8, 4, 2>(A, V, K, V, B, out, 2));
        vs.push_back(mkct32<CT, 2, 4, 8>(A, V, K, V, B, out, 2));
        vs.push_back(mkct32<CT, 4, 4, 8>(A, V, K, V, B, out, 2));
        vs.push_back(mkct32<CT, 4, 4, 8>(A, V, K, V, B, out, 1));
        vs.push_back(mkct32<CT, 4, 4, 4>(A, V, K, V, B, out, 2));
        vs.push_back(mkct32<CT, 4, 2, 8>(A, V, K, V, B, out, 4));
        vs.push_back(mkct32<CT, 8, 4, 8>(A, V, K, V, B, out, 1));
    } else {
        vs.push_back(mkct32<CT, 2, 4, 8>(A, V, K, V, B, out, 2));
        vs.push_back(mkct32<CT, 4, 4, 8>(A, V, K, V, B, out, 1));
        vs.push_back(mkct32<CT, 2, 4, 4>(A, V, K, V, B, out, 2));
    }
    for (size_t k = 1; k < vs.size(); ++k) check<T>(name, vs[k], vs[0], out, V, 16 * CT, 2e-5);
    if (!small) bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(which, "all");
    if (all || !strcmp(which, "odd")) {
        suite<4>("odd_m64", 2048, 2048, true);
        suite<8>("odd_m128", 1024, 3072, true);
        suite<2>("odd_m32", 1024, 6656, true);
    }
    if (all || !strcmp(which, "c3")) {
        suite<4>("c3l_xty", 50048, 20480, false);      // 20480 = 40 x 512: whole super tiles for every variant
        suite<4>("c3l_xw", 20480, 50176, false);       // 50176 = 98 x 512
    }
    if (!strcmp(which, "c3hint")) {
        typedef float T;
        const int64_t K = 50048, V = 20480;
        T *A, *B, *out;
        CK(hipMalloc(&A, sizeof(T) * K * V)); CK(hipMalloc(&B, sizeof(T) * K * 64)); CK(hipMalloc(&out, sizeof(T) * 48 * V * 64));
        {
            std::vector<T> h((size_t)K * V);
            for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
            CK(hipMemcpy(A, h.data(), sizeof(T) * K * V, hipMemcpyHostToDevice));
            CK(hipMemcpy(B, h.data(), sizeof(T) * K * 64, hipMemcpyHostToDevice));
        }
        printf("== c3l_xty scheduler hints: K=%ld V=%ld Mp=64 float32\n", (long)K, (long)V);
        std::vector<Variant> vs;
        vs.push_back(mkct<T, 4, 4, 4, 4, true, 0>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, 4, 4, 4, 4, true, 3>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, 4, 4, 4, 4, true, 4>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, 4, 4, 4, 8, true, 0>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, 4, 4, 4, 8, true, 3>(A, V, K, V, B, out, 2));
        vs.push_back(mkct<T, 4, 4, 4, 4, true, 1>(A, V, K, V, B, out, 2));
        vs.push_back(mkct3<T, 4, 4, 4, 4>(A, V, K, V, B, out, 2));
        vs.push_back(mkct3<T, 4, 4, 4, 4>(A, V, K, V, B, out, 3));
        vs.push_back(mkct3<T, 4, 4, 4, 2>(A, V, K, V, B, out, 2));
        for (size_t k = 1; k < vs.size(); ++k) check<T>("hint", vs[k], vs[0], out, V, 64, 2e-5);
        bench(vs, sizeof(T) * ((double)K * V + 64.0 * (K + V)) / 1e9, 2.0 * K * V * 64 / 1e12, 7, 5);
    }
    if (!strcmp(which, "c3zero")) {
        suite<4>("c3l_xty_zero_X", 50048, 20480, false, true);
    }
    if (all || !strcmp(which, "c4")) {
        suite<8>("c4l_xty", 50048, 20480, false);
    }
    return 0;
}
