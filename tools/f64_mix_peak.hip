// f64_mix_peak.hip - do v_mfma_f64_4x4x4 and v_fma_f64 run side by side on one SIMD?  (GPU box only; not part of the product)
//   same-wave: NM MFMAs + NV FMAs per loop trip, independent accumulators;  split: waves 0-3 MFMA only, waves 4-7 FMA only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NM, int NV>
__global__ void __launch_bounds__(256) k_same(double* out, int iters, double a0, double b0, const double* __restrict__ bsrc) {
    double am[NM > 0 ? NM : 1], av[NV > 0 ? NV : 1];
    double a[4], x[4];
    for (int i = 0; i < NM; ++i) am[i] = 0.0;
    for (int i = 0; i < NV; ++i) av[i] = 0.0;
    for (int i = 0; i < 4; ++i) { a[i] = a0 + i * 0.01 + threadIdx.x * 1e-9; x[i] = b0 + i * 0.03 + threadIdx.x * 1e-9; }
    double bs[8];
    for (int i = 0; i < 8; ++i) bs[i] = bsrc[i];            // uniform: lives in SGPRs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (NM > NV ? NM : NV); ++i) {
            if (i < NM) am[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], x[(i + 1) & 3], am[i], 0, 0, 0);
            if (i < NV) av[i] = __builtin_fma(x[i & 3], bs[i & 7], av[i]);
        }
    }
    double s = 0;
    for (int i = 0; i < NM; ++i) s += am[i];
    for (int i = 0; i < NV; ++i) s += av[i];
    if (s == 12345.678) out[0] = s;
}

template <int NM, int NV>
__global__ void __launch_bounds__(512) k_split(double* out, int iters_m, int iters_v, double a0, double b0, const double* __restrict__ bsrc) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        double am[NM], a[4], x[4];
        for (int i = 0; i < NM; ++i) am[i] = 0.0;
        for (int i = 0; i < 4; ++i) { a[i] = a0 + i * 0.01 + threadIdx.x * 1e-9; x[i] = b0 + i * 0.03; }
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) am[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], x[(i + 1) & 3], am[i], 0, 0, 0);
        }
        for (int i = 0; i < NM; ++i) s += am[i];
    } else {
        double av[NV], x[4], bs[8];
        for (int i = 0; i < NV; ++i) av[i] = 0.0;
        for (int i = 0; i < 4; ++i) x[i] = b0 + i * 0.03 + threadIdx.x * 1e-9;
        for (int i = 0; i < 8; ++i) bs[i] = bsrc[i];
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) av[i] = __builtin_fma(x[i & 3], bs[i & 7], av[i]);
        }
        for (int i = 0; i < NV; ++i) s += av[i];
    }
    if (s == 12345.678) out[0] = s;
}

template <typename F> double run(const char* tag, F launch, double flop) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%-64s %8.1f us  %6.1f TF/s\n", tag, ms * 1e3, flop / ms / 1e9);
    return ms;
}

int main() {
    double *o64, *bs;
    CK(hipMalloc(&o64, 64)); CK(hipMalloc(&bs, 64));
    double hb[8] = {0.3, -0.2, 0.11, 0.7, -0.9, 0.4, 0.25, -0.6};
    CK(hipMemcpy(bs, hb, 64, hipMemcpyHostToDevice));
    const int iters = 2000;
    const double MF = 512.0, VF = 128.0;                    // flops per wave instruction: 4 blocks 4x4x4 x 2 = 512; 64 lanes x 2 = 128
    for (int blocks : {512, 1024}) {
        const double waves = blocks * 4.0;
        char tag[128];
#define SAME(NM, NV)                                                                                               \
        snprintf(tag, 128, "same wave: %2d mfma + %2d fma per trip, %d blocks x 4 waves", NM, NV, blocks);       \
        run(tag, [&] { hipLaunchKernelGGL((k_same<NM, NV>), dim3(blocks), dim3(256), 0, 0, o64, iters, 0.73, -1.21, bs); }, \
            waves * iters * (NM * MF + NV * VF));
        SAME(16, 0) SAME(0, 32) SAME(0, 64) SAME(16, 16) SAME(16, 32) SAME(16, 48) SAME(16, 64) SAME(8, 32) SAME(8, 64)
        // split: 4 MFMA waves (16 per trip) + 4 FMA waves (64 per trip: the same 4x17=68 vs 64x4=256 cycles -> scale trips)
        snprintf(tag, 128, "split: waves 0-3 16 mfma x %d trips only, %d blocks x 8 waves", iters, blocks);
        run(tag, [&] { hipLaunchKernelGGL((k_split<16, 64>), dim3(blocks), dim3(512), 0, 0, o64, iters, 0, 0.73, -1.21, bs); }, waves * iters * 16 * MF);
        snprintf(tag, 128, "split: waves 4-7 64 fma x %d trips only", iters / 4);
        run(tag, [&] { hipLaunchKernelGGL((k_split<16, 64>), dim3(blocks), dim3(512), 0, 0, o64, 0, iters / 4, 0.73, -1.21, bs); }, waves * (iters / 4) * 64 * VF);
        snprintf(tag, 128, "split: both (16 mfma x %d | 64 fma x %d)", iters, iters / 4);
        run(tag, [&] { hipLaunchKernelGGL((k_split<16, 64>), dim3(blocks), dim3(512), 0, 0, o64, iters, iters / 4, 0.73, -1.21, bs); },
            waves * (iters * 16 * MF + (iters / 4) * 64 * VF));
    }
    return 0;
}
