// gemm_probe7.hip - float32 small-shard pass (config-2 shape in float32): ablation of gemm_tn (GPU box only)
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
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };
template <int CT, int RT, int KW, int MODE, int U, bool NTA>
Variant mk(const float* A, int64_t lda, int64_t K, int64_t vcols, const float* B, float* out, int S) {
    auto kern = gemm_tn_probe_kernel<float, CT, RT, KW, false, MODE, U, NTA>;
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * 4;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    char buf[200];
    snprintf(buf, 200, "tn f32 RT=%d KW=%d U=%d mode=%d NT=%d S=%d blocks=%d bpc=%d", RT, KW, U, MODE, (int)NTA, S, (int)(vcols / (16 * RT)) * S, bpc);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, (int64_t)(16 * RT), B, (const float*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}};
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
void suite(const char* name, int64_t K, int64_t V) {
    const int Mp = 32;
    float *A, *B, *out;
    CK(hipMalloc(&A, 4 * K * V)); CK(hipMalloc(&B, 4 * K * Mp)); CK(hipMalloc(&out, 4 * 40 * V * Mp));
    std::vector<float> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(A, h.data(), 4 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 7, 4 * K * Mp, hipMemcpyHostToDevice));
    const double gb = 4 * ((double)K * V + (double)Mp * (K + V)) / 1e9, tf = 2.0 * K * V * Mp / 1e12;
    printf("== %s: K=%ld V=%ld Mp=32 f32\n", name, (long)K, (long)V);
    std::vector<Variant> vs;
    for (int S : {3, 6, 12}) {
        vs.push_back(mk<2, 4, 4, 0, 4, false>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 4, 4, 0, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 4, 4, 1, 4, false>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 4, 4, 1, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 4, 4, 2, 4, false>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 4, 4, 0, 8, false>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 4, 8, 0, 4, false>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 8, 4, 0, 4, false>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 8, 4, 0, 4, true>(A, V, K, V, B, out, S));
        vs.push_back(mk<2, 8, 4, 1, 4, true>(A, V, K, V, B, out, S));
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}
int main() {
    suite("c2f32_xty", 10048, 5120);
    return 0;
}
