// gemm_probe.hip - ablation / geometry probe for gemm_tn_kernel (GPU box only; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_probe tools/gemm_probe.hip
//   tools/gemm_probe          (C2 shape: K=10048 rows x 5056 cols f64, Mp=32; and the XT orientation)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include "probe_kernels.hpp"
using namespace lcx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int CT, int RT, int KW, int MODE, int U>
void run(const char* tag, const T* A, int64_t lda_in, int64_t K, int64_t vcols, const T* B, T* out, int S, double gbytes, double tflop, bool panel = false) {
    const int64_t lda = panel ? 16 * RT : lda_in, tstride = panel ? K * 16 * RT : 16 * RT;
    dim3 grid((unsigned)(vcols / (16 * RT)), (unsigned)S);
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    auto kern = gemm_tn_probe_kernel<T, CT, RT, KW, false, MODE, U>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * KW), lds, 0, A, lda, tstride, B, (const T*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr);
    CK(hipEventRecord(a, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(64 * KW), lds, 0, A, lda, tstride, B, (const T*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= iters;
    printf("%-10s %s RT=%d KW=%d U=%d S=%2d mode=%d blocks=%5u bpc=%d : %7.1f us  %6.0f GB/s  %5.1f TF/s\n", tag, panel ? "panel" : "plain", RT, KW, U, S, MODE,
           grid.x * grid.y, bpc, ms * 1e3, gbytes / ms * 1e3, tflop / ms * 1e3);
    fflush(stdout);
}

template <typename T, int CT>
void suite(const char* name, int64_t K, int64_t V) {
    // A: [K][lda=V], contraction over K rows, output tiles over V columns
    T *A, *B, *out;
    CK(hipMalloc(&A, sizeof(T) * K * V));
    CK(hipMalloc(&B, sizeof(T) * K * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 32 * V * 16 * CT));
    std::vector<T> h((size_t)K * V);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (T)((double)rand() / RAND_MAX - 0.5);
    CK(hipMemcpy(A, h.data(), sizeof(T) * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), sizeof(T) * K * 16 * CT, hipMemcpyHostToDevice));
    const double gb = sizeof(T) * ((double)K * V + 16.0 * CT * (K + V)) / 1e9, tf = 2.0 * K * V * 16 * CT / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d elt=%zu\n", name, (long)K, (long)V, 16 * CT, sizeof(T));
    const int tiles2 = (int)(V / 32), tiles4 = (int)(V / 64);
    (void)tiles2; (void)tiles4;
#define R(RT, KW, MODE, U, S) run<T, CT, RT, KW, MODE, U>(name, A, V, K, V, B, out, S, gb, tf, false)
#define PN(RT, KW, MODE, U, S) run<T, CT, RT, KW, MODE, U>(name, A, V, K, V, B, out, S, gb, tf, true)
    for (int S : {1, 2, 3, 4, 6, 8}) R(2, 8, 0, 4, S);
    for (int S : {1, 2, 3, 4, 6, 8}) PN(2, 8, 0, 4, S);
    for (int S : {1, 2, 3, 4, 6, 8}) PN(2, 8, 1, 4, S);
    for (int S : {1, 2, 3, 4, 6, 8}) PN(2, 4, 0, 4, S);
    for (int S : {1, 2, 3, 4, 6, 8}) PN(2, 8, 0, 8, S);
    for (int S : {1, 2, 3, 4, 6, 8}) PN(2, 8, 1, 8, S);
    for (int S : {1, 2, 3, 4, 6, 8, 12}) PN(4, 8, 0, 4, S);
    for (int S : {1, 2, 3, 4, 6, 8, 12}) PN(4, 8, 1, 4, S);
    for (int S : {2, 3, 4, 6, 8, 12}) PN(4, 4, 0, 4, S);
    for (int S : {2, 3, 4, 6, 8, 12}) PN(4, 4, 0, 8, S);
    for (int S : {2, 3, 4, 6, 8, 12}) PN(4, 4, 1, 8, S);
#undef PN
#undef R
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "c2";
    if (!strcmp(which, "c2")) {
        suite<double, 2>("c2_xty", 10048, 5056);   // X^T.Y : contraction over samples
        suite<double, 2>("c2_xw", 5056, 10048);    // X.W^T via XT : contraction over variables
    } else if (!strcmp(which, "c3")) {
        suite<float, 4>("c3l_xty", 50048, 20032);
        suite<float, 4>("c3l_xw", 20032, 50048);
    }
    return 0;
}
