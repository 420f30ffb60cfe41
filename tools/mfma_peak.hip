// mfma_peak.hip - issue-rate ceilings of the MFMA shapes the fit path uses (GPU box only; not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_f64(double* out, int iters, double a0, double b0, long long* clk) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (clk && blockIdx.x == 3 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f32(float* out, int iters, float a0, float b0, long long* clk) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    if (clk && blockIdx.x == 3 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

// distinct A / B registers per accumulator (a real kernel never feeds one register pair to every MFMA)
template <int NACC>
__global__ void __launch_bounds__(256) k_f64_var(double* out, int iters, double a0, double b0, long long* clk) {
    d4 acc[NACC];
    double a[NACC], b[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = (d4){0, 0, 0, 0}; a[i] = a0 + i * 0.01 + threadIdx.x * 1e-9; b[i] = b0 - i * 0.02; }
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + 1) % NACC], acc[i], 0, 0, 0);
    }
    if (clk && blockIdx.x == 3 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f64_4x4_var(double* out, int iters, double a0, double b0, long long* clk) {
    double acc[NACC], a[4], b[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = 0.0; b[i] = b0 - i * 0.02; }
    for (int i = 0; i < 4; ++i) a[i] = a0 + i * 0.01 + threadIdx.x * 1e-9;
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[i], acc[i], 0, 0, 0);
    }
    if (clk && blockIdx.x == 3 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_f64_4x4(double* out, int iters, double a0, double b0, long long* clk) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
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
    printf("%-44s %8.1f us  %6.1f TF/s  shader clk %4.0f MHz\n", tag, ms * 1e3, flop / ms / 1e9, (double)hc[0] / hc[1] * 100.0);
}

int main() {
    double* o64; float* o32; long long* clk;
    CK(hipMalloc(&o64, 64)); CK(hipMalloc(&o32, 64)); CK(hipMalloc(&clk, 64));
    const int iters = 2000;
    for (int blocks : {256, 512, 1024}) {
        const double waves = blocks * 4.0;
        char tag[96];
        snprintf(tag, 96, "f64 16x16x4, 8 acc, %d blocks x 4 waves (1.0)", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f64<8>, dim3(blocks), dim3(256), 0, 0, o64, iters, 1.0, 1.0, clk); }, waves * iters * 8 * 2048.0, clk);
        snprintf(tag, 96, "f64 16x16x4, 8 acc, %d blocks (random-ish)", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f64<8>, dim3(blocks), dim3(256), 0, 0, o64, iters, 0.7312345, -1.218765, clk); }, waves * iters * 8 * 2048.0, clk);
        snprintf(tag, 96, "f64 16x16x4, 4 acc, %d blocks", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f64<4>, dim3(blocks), dim3(256), 0, 0, o64, iters, 0.7312345, -1.218765, clk); }, waves * iters * 4 * 2048.0, clk);
        snprintf(tag, 96, "f64 16x16x4, 8 acc, distinct operands, %d blocks", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f64_var<8>, dim3(blocks), dim3(256), 0, 0, o64, iters, 0.7312345, -1.218765, clk); }, waves * iters * 8 * 2048.0, clk);
        snprintf(tag, 96, "f64 4x4x4, 16 acc, distinct operands, %d blocks", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f64_4x4_var<16>, dim3(blocks), dim3(256), 0, 0, o64, iters, 0.7312345, -1.218765, clk); }, waves * iters * 16 * 512.0, clk);
        snprintf(tag, 96, "f64 4x4x4 (4 blocks), 16 acc, %d blocks", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f64_4x4<16>, dim3(blocks), dim3(256), 0, 0, o64, iters, 0.7312345, -1.218765, clk); }, waves * iters * 16 * 512.0, clk);
        snprintf(tag, 96, "f32 16x16x4, 16 acc, %d blocks", blocks);
        run(tag, [&] { hipLaunchKernelGGL(k_f32<16>, dim3(blocks), dim3(256), 0, 0, o32, iters, 0.7312345f, -1.218765f, clk); }, waves * iters * 16 * 2048.0, clk);
    }
    return 0;
}
