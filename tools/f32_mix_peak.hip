// f32_mix_peak.hip - do v_mfma_f32_16x16x4 and v_fma_f32 / v_pk_fma_f32 add up on one SIMD?  (GPU box only; not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NM, int NV, bool PK>
__global__ void __launch_bounds__(256) k_same(float* out, int iters, float a0, float b0, const float* __restrict__ bsrc, long long* clk) {
    f4 am[NM > 0 ? NM : 1];
    f2 av[NV > 0 ? NV : 1];
    float a[4], x[4];
    for (int i = 0; i < NM; ++i) am[i] = (f4){0, 0, 0, 0};
    for (int i = 0; i < NV; ++i) av[i] = (f2){0, 0};
    for (int i = 0; i < 4; ++i) { a[i] = a0 + i * 0.01f + threadIdx.x * 1e-4f; x[i] = b0 + i * 0.03f + threadIdx.x * 1e-4f; }
    float bs[8];
    for (int i = 0; i < 8; ++i) bs[i] = bsrc[i];
    long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (NM > NV ? NM : NV); ++i) {
            if (i < NM) am[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], x[(i + 1) & 3], am[i], 0, 0, 0);
            if (i < NV) {
                if (PK) av[i] = __builtin_elementwise_fma((f2){x[i & 3], x[(i + 2) & 3]}, (f2){bs[i & 7], bs[i & 7]}, av[i]);
                else av[i].x = __builtin_fmaf(x[i & 3], bs[i & 7], av[i].x);
            }
        }
    }
    if (clk && blockIdx.x == 3 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
    float s = 0;
    for (int i = 0; i < NM; ++i) s += am[i][0] + am[i][1] + am[i][2] + am[i][3];
    for (int i = 0; i < NV; ++i) s += av[i].x + av[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <typename F> double run(const char* tag, F launch, double flop, long long* clk) {
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
    printf("%-64s %8.1f us  %6.1f TF/s  shader clk %4.0f MHz\n", tag, ms * 1e3, flop / ms / 1e9, (double)hc[0] / hc[1] * 100.0);
    return ms;
}

int main() {
    float *o, *bs; long long* clk;
    CK(hipMalloc(&o, 64)); CK(hipMalloc(&bs, 64)); CK(hipMalloc(&clk, 64));
    float hb[8] = {0.3f, -0.2f, 0.11f, 0.7f, -0.9f, 0.4f, 0.25f, -0.6f};
    CK(hipMemcpy(bs, hb, 32, hipMemcpyHostToDevice));
    const int iters = 20000;
    const double MF = 2048.0;                               // 16x16x4 x 2
    for (int blocks : {1024}) {
        const double waves = blocks * 4.0;
        char tag[128];
#define SAME(NM, NV, PK)                                                                                           \
        snprintf(tag, 128, "same wave: %2d mfma f32 16x16x4 + %2d %s per trip, %d blocks", NM, NV, PK ? "v_pk_fma_f32" : "v_fma_f32", blocks); \
        run(tag, [&] { hipLaunchKernelGGL((k_same<NM, NV, PK>), dim3(blocks), dim3(256), 0, 0, o, iters, 0.73f, -1.21f, bs, clk); }, \
            waves * iters * (NM * MF + NV * (PK ? 256.0 : 128.0)), clk);
        SAME(16, 0, false) SAME(16, 8, false) SAME(16, 0, false) SAME(16, 4, false) SAME(16, 2, false) SAME(16, 0, false)
        SAME(0, 32, false) SAME(0, 32, true)
        SAME(16, 8, false) SAME(16, 16, false) SAME(16, 32, false) SAME(16, 64, false)
        SAME(16, 8, true) SAME(16, 16, true) SAME(16, 32, true) SAME(16, 0, false)
    }
    return 0;
}
