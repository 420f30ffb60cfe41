// read_probe.hip - what does a read-only stream of a 400 MB matrix reach on MI355X, by access shape?
// (GPU box only; not part of the product.)   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/read_probe tools/read_probe.hip
//
// Variants:
//   linear   : every wave instruction reads 1 KiB contiguous (lane*16 B), UNR loads in flight per lane
//   rows     : every wave instruction reads 4 rows x 256 B (the gemm_tn pattern: 16 lanes x 16 B per row),
//              row stride = lda, a wave walks down the rows of its 256/512-B column strip
//   rows32   : lane i reads 32 B at i*32 as two 16 B loads (the f64 RT=4 pattern before the fix)
//   NT       : __builtin_nontemporal_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

template <int UNR, bool NT>
__global__ void __launch_bounds__(256) linear_kernel(const f4* __restrict__ a, int64_t n16, float* __restrict__ out) {
    f4 acc = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += ld<NT>(a + i);
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// matrix [K rows][lda bytes-of-16]; a wave owns a strip of W16 16-B pieces per row per lane-group;
// block = KW waves splitting the rows; grid = (strips, S)
template <int U, bool NT, int PIECES /* 16-B loads per lane per row */, bool INTERLEAVED>
__global__ void __launch_bounds__(256) rows_kernel(const f4* __restrict__ a, int64_t lda16, int K, int S, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int part = blockIdx.y * 4 + wave, nparts = S * 4;
    const int ng = K / (4 * U);
    const int g0 = (int)((int64_t)ng * part / nparts), g1 = (int)((int64_t)ng * (part + 1) / nparts);
    // strip of 16*PIECES pieces per row
    const f4* base = a + (int64_t)blockIdx.x * (16 * PIECES) + (int64_t)q * lda16;
    f4 acc = {0, 0, 0, 0};
    for (int g = g0; g < g1; ++g) {
        f4 v[U][PIECES];
#pragma unroll
        for (int st = 0; st < U; ++st)
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const int col = INTERLEAVED ? (i * PIECES + p) : (p * 16 + i);
                v[st][p] = ld<NT>(base + ((int64_t)g * 4 * U + 4 * st) * lda16 + col);
            }
#pragma unroll
        for (int st = 0; st < U; ++st)
#pragma unroll
            for (int p = 0; p < PIECES; ++p) acc += v[st][p];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <typename F> static void timeit(const char* tag, double bytes, F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    printf("%-44s %8.1f us  %6.0f GB/s\n", tag, ms * 1e3, bytes / ms / 1e6);
    fflush(stdout);
}

int main() {
    const int64_t K = 10048, V = 5056;                 // f64 elements -> row = 40448 B = 2528 pieces
    const int64_t lda16 = V * 8 / 16, n16 = K * lda16;
    const double bytes = (double)n16 * 16;
    f4* A; float* out;
    CK(hipMalloc(&A, n16 * 16)); CK(hipMalloc(&out, 64));
    CK(hipMemset(A, 0x3c, n16 * 16));
    char tag[128];
    for (int blocks : {512, 1024, 2048, 4096}) {
        snprintf(tag, 128, "linear UNR=4 blocks=%d", blocks);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((linear_kernel<4, false>), dim3(blocks), dim3(256), 0, 0, A, n16, out); });
        snprintf(tag, 128, "linear UNR=8 blocks=%d", blocks);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((linear_kernel<8, false>), dim3(blocks), dim3(256), 0, 0, A, n16, out); });
        snprintf(tag, 128, "linear UNR=8 NT blocks=%d", blocks);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((linear_kernel<8, true>), dim3(blocks), dim3(256), 0, 0, A, n16, out); });
        snprintf(tag, 128, "linear UNR=16 blocks=%d", blocks);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((linear_kernel<16, false>), dim3(blocks), dim3(256), 0, 0, A, n16, out); });
    }
    // strips: PIECES=2 -> 512 B per row per wave -> 79 strips ; PIECES=1 -> 158 strips
    for (int S : {3, 6, 9, 12, 13, 19, 26}) {
        snprintf(tag, 128, "rows 512B contiguous-halves U=4 S=%d blocks=%d", S, 79 * S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<4, false, 2, false>), dim3(79, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
        snprintf(tag, 128, "rows 512B interleaved(32B/lane) U=4 S=%d", S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<4, false, 2, true>), dim3(79, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
        snprintf(tag, 128, "rows 512B contiguous-halves U=8 S=%d", S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<8, false, 2, false>), dim3(79, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
        snprintf(tag, 128, "rows 512B contiguous-halves U=8 NT S=%d", S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<8, true, 2, false>), dim3(79, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
        snprintf(tag, 128, "rows 1024B contiguous U=4 S=%d blocks=%d", S, 39 * S);
        timeit(tag, bytes * 39 * 4 / (79 * 2), [&] { hipLaunchKernelGGL((rows_kernel<4, false, 4, false>), dim3(39, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
    }
    for (int S : {2, 3, 4, 6, 13}) {
        snprintf(tag, 128, "rows 256B U=4 S=%d blocks=%d", S, 158 * S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<4, false, 1, false>), dim3(158, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
        snprintf(tag, 128, "rows 256B U=8 S=%d", S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<8, false, 1, false>), dim3(158, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
        snprintf(tag, 128, "rows 256B U=16 S=%d", S);
        timeit(tag, bytes, [&] { hipLaunchKernelGGL((rows_kernel<16, false, 1, false>), dim3(158, S), dim3(256), 0, 0, A, lda16, (int)K, S, out); });
    }
    return 0;
}
