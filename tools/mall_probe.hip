// mall_probe.hip - does the 256 MB Infinity Cache serve a re-read of part of a streamed matrix on MI355X?
// (GPU box only; not part of the product.)   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mall_probe tools/mall_probe.hip
//
// A read-only stream over a buffer of S MB is launched back to back, either always front-to-back ("fwd") or
// alternating front-to-back / back-to-front ("zigzag": what the next launch reads first is what the previous one
// read last).  If the last-level cache keeps the most recently read lines, zigzag turns min(S, cache)/S of every
// pass into cache hits.  Reported: microseconds per pass and the effective GB/s.
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
__global__ void __launch_bounds__(256) stream_kernel(const f4* __restrict__ a, int64_t n16, int reverse, float* __restrict__ out) {
    f4 acc = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t steps = n16 / (stride * UNR);
    const int64_t lane0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t s = 0; s < steps; ++s) {
        const int64_t ss = reverse ? steps - 1 - s : s;
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = ld<NT>(a + (ss * UNR + u) * stride + lane0);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <bool NT>
static void run(const f4* A, int64_t mb, bool zigzag, float* out) {
    const int blocks = 512;
    const int64_t quantum = (int64_t)blocks * 256 * 4;             // f4 per step
    int64_t n16 = mb * 1000000 / 16 / quantum * quantum;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int dir = 0;
    for (int i = 0; i < 4; ++i) { hipLaunchKernelGGL((stream_kernel<4, NT>), dim3(blocks), dim3(256), 0, 0, A, n16, dir, out); if (zigzag) dir ^= 1; }
    CK(hipEventRecord(a, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL((stream_kernel<4, NT>), dim3(blocks), dim3(256), 0, 0, A, n16, dir, out); if (zigzag) dir ^= 1; }
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    printf("%5ld MB %-7s %-3s %8.1f us/pass %7.0f GB/s\n", (long)mb, zigzag ? "zigzag" : "fwd", NT ? "nt" : "", ms * 1e3, (double)n16 * 16 / ms / 1e6);
    fflush(stdout);
}

int main() {
    const int64_t maxmb = 1000;
    f4* A; float* out;
    CK(hipMalloc(&A, maxmb * 1000000)); CK(hipMalloc(&out, 64));
    CK(hipMemset(A, 0x3c, maxmb * 1000000));
    for (int64_t mb : {16, 32, 64, 100, 128, 160, 200, 256, 320, 404, 600, 808}) {
        run<false>(A, mb, false, out);
        run<false>(A, mb, true, out);
        run<true>(A, mb, false, out);
        run<true>(A, mb, true, out);
    }
    return 0;
}
