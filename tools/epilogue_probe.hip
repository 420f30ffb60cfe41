// epilogue_probe.hip - where do the microseconds of moments_epilogue_kernel go?  (GPU box only; not part of the product.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I linearcorex_amd/csrc -o tools/epilogue_probe tools/epilogue_probe.hip
// Back-to-back launches of the production kernel and of its ablation variants (template parameter ABL) on random
// data at the config-2 (V = 5000, 6 slots) and config-5 (V = 20000, 3 slots) shapes, float64, 32 padded factors.
// profiles/r01_epilogue_probe_tail.txt is the output of the version that still had the fused tail (ticket, last block
// sums and publishes): 19.5 us with it, 8.1 us without at config 2 - which is why the tail moved into gram_tc_kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include "probe_kernels.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace lcx;

template <int ABL>
static float run(int grid, int iters, const double* dpart, int nsplit, int64_t pstride, double* d_out, const double* W, const double* ry,
                 int64_t V, double* rho, double* rir, double* qij, double* si, double* q2, double* hs, double* tcpart,
                 unsigned int* ticket, double* sbuf, SetState* st, SetState* host) {
    constexpr int Mp = 32, VPB = PV_THREADS / Mp;
    const size_t lds = ((size_t)Mp * Mp + (size_t)VPB * Mp) * sizeof(double);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned int seq = 0;
    auto launch = [&]() {
        hipLaunchKernelGGL((moments_epilogue_probe_kernel<double, Mp, ABL>), dim3(grid), dim3(PV_THREADS), lds, 0, dpart, nsplit, pstride,
                           (const double*)nullptr, (const double*)nullptr, 0.0, d_out, W, ry, V, 10000.0, 0.1, rho, rir, qij, si, q2, hs,
                           tcpart, (const int*)nullptr);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

static void suite(const char* name, int64_t V, int nsplit) {
    constexpr int Mp = 32;
    const int64_t ldx = (V + 63) / 64 * 64, n = ldx * Mp;
    double *dpart, *d_out, *W, *ry, *rho, *rir, *qij, *si, *q2, *hs, *tcpart, *sbuf;
    unsigned int* ticket; SetState *st, *host;
    CK(hipMalloc(&dpart, sizeof(double) * n * nsplit)); CK(hipMalloc(&d_out, sizeof(double) * n)); CK(hipMalloc(&W, sizeof(double) * n));
    CK(hipMalloc(&ry, sizeof(double) * Mp * Mp)); CK(hipMalloc(&rho, sizeof(double) * n)); CK(hipMalloc(&rir, sizeof(double) * n));
    CK(hipMalloc(&qij, sizeof(double) * n)); CK(hipMalloc(&si, sizeof(double) * ldx)); CK(hipMalloc(&q2, sizeof(double) * ldx));
    CK(hipMalloc(&hs, sizeof(double) * ldx)); CK(hipMalloc(&tcpart, sizeof(double) * 2 * 4096)); CK(hipMalloc(&sbuf, sizeof(double) * 4096));
    CK(hipMalloc(&ticket, 256)); CK(hipMemset(ticket, 0, 256)); CK(hipMalloc(&st, sizeof(SetState) * 2));
    CK(hipMemset(st, 0, sizeof(SetState) * 2));
    CK(hipHostMalloc((void**)&host, sizeof(SetState) * 2, hipHostMallocMapped | hipHostMallocCoherent));
    std::vector<double> h(n * nsplit);
    for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * 100.0;      // rho ~ d / N stays well inside (-1, 1)
    CK(hipMemcpy(dpart, h.data(), sizeof(double) * n * nsplit, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < n; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * 0.1;
    CK(hipMemcpy(W, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    for (int i = 0; i < Mp * Mp; ++i) h[i] = (i / Mp == i % Mp) ? 1.0 : 0.01;
    CK(hipMemcpy(ry, h.data(), sizeof(double) * Mp * Mp, hipMemcpyHostToDevice));
    const int64_t groups = (V + 7) / 8;
    printf("== %s: V=%ld Mp=32 slots=%d f64 (groups of 8 variables: %ld)\n", name, (long)V, nsplit, (long)groups);
#define RUN(ABL, GRID, TAG) printf("  grid %5d  %-46s %7.1f us\n", (int)(GRID), TAG, run<ABL>((int)(GRID), 30, dpart, nsplit, n, d_out, W, ry, V, rho, rir, qij, si, q2, hs, tcpart, ticket, sbuf, st, host)); fflush(stdout);
    const int g0 = (int)(groups < 1024 ? groups : 1024);
    RUN(0, g0, "production");
    RUN(1, g0, "no m x m matvec");
    RUN(2, g0, "no M x V stores");
    RUN(8, g0, "no logarithms");
    RUN(16, g0, "first slot only");
    RUN(27, g0, "all of the above (loads + rho + reductions)");
    for (int g : {256, 512, 768, 1024, 1536, 2048}) {
        if (g > groups) break;
        RUN(0, g, "production, other grid");
    }
#undef RUN
}

int main() {
    suite("config 2", 5000, 6);
    suite("config 5", 20000, 3);
    return 0;
}
