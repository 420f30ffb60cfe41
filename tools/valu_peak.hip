// valu_peak.hip - v_fma_f64 issue-rate ceiling (VGPR and SGPR multiplicand) vs the f64 MFMA ceiling (GPU box only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC, bool SGPR>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, const double* __restrict__ bsrc, long long* clk) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = 0.7312345 + threadIdx.x * 1e-9;
    double b[8];
    for (int i = 0; i < 8; ++i) b[i] = SGPR ? bsrc[i] : bsrc[i] + threadIdx.x * 1e-12;   // uniform -> SGPRs when SGPR
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, b[i & 7], acc[i]);
        a += 1e-9;
    }
    if (clk && blockIdx.x == 3 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

template <typename F> void run(const char* tag, F launch, double flop, long long* clk) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    long long hc[2];
    CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    printf("%-52s %8.1f us  %6.1f TF/s  shader clk %4.0f MHz\n", tag, ms * 1e3, flop / ms / 1e9, (double)hc[0] / hc[1] * 100.0);
}

int main() {
    double *o, *b; long long* clk;
    CK(hipMalloc(&o, 64)); CK(hipMalloc(&b, 64)); CK(hipMalloc(&clk, 64));
    double hb[8] = {1.01, -0.99, 0.5, 0.25, -0.75, 1.5, -1.25, 0.125};
    CK(hipMemcpy(b, hb, 64, hipMemcpyHostToDevice));
    const int iters = 4000;
    for (int blocks : {256, 512, 1024, 2048}) {
        const double thr = blocks * 256.0;
        char tag[96];
        snprintf(tag, 96, "v_fma_f64 VGPR b, 32 acc, %d blocks x 256", blocks);
        run(tag, [&] { hipLaunchKernelGGL((k_fma<32, false>), dim3(blocks), dim3(256), 0, 0, o, iters, b, clk); }, thr * iters * 32 * 2.0, clk);
        snprintf(tag, 96, "v_fma_f64 SGPR b, 32 acc, %d blocks x 256", blocks);
        run(tag, [&] { hipLaunchKernelGGL((k_fma<32, true>), dim3(blocks), dim3(256), 0, 0, o, iters, b, clk); }, thr * iters * 32 * 2.0, clk);
        snprintf(tag, 96, "v_fma_f64 SGPR b, 64 acc, %d blocks x 256", blocks);
        run(tag, [&] { hipLaunchKernelGGL((k_fma<64, true>), dim3(blocks), dim3(256), 0, 0, o, iters, b, clk); }, thr * iters * 64 * 2.0, clk);
    }
    return 0;
}
