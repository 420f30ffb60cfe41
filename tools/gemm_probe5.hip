// gemm_probe5.hip - float64 contraction on the VECTOR pipe: v_fma_f64 with the small operand in SGPRs (GPU box only).
// MI355X measured ceilings (tools/mfma_peak, tools/valu_peak): v_mfma_f64_16x16x4 47.6 TF/s, v_fma_f64 62-72 TF/s.
//   D[v][j] = sum_n A[n][v] B[n][j]: lane = 2 adjacent columns v (16-byte loads, 1 KiB per wave and row),
//   B[n][0..Mp) is wave-uniform -> scalar loads, acc[2][Mp] per lane.
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
typedef double d2 __attribute__((ext_vector_type(2)));

template <int Mp, int KW, int R, bool NT>
__global__ void __launch_bounds__(64 * KW)
vf(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, double* __restrict__ out, int64_t out_rows, int K, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);          // [128][Mp]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t v0 = (int64_t)blockIdx.x * 128;
    const int part = blockIdx.y * KW + wave, nparts = nsplit * KW;
    const int k0 = (int)((int64_t)K * part / nparts), k1 = (int)((int64_t)K * (part + 1) / nparts);
    double acc0[Mp], acc1[Mp];
#pragma unroll
    for (int j = 0; j < Mp; ++j) { acc0[j] = 0.0; acc1[j] = 0.0; }
    const d2* ap = reinterpret_cast<const d2*>(A + v0 + 2 * lane);
    const int64_t lda2 = lda / 2;
    d2 a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) if (k0 + r < k1) a[r] = NT ? __builtin_nontemporal_load(ap + (int64_t)(k0 + r) * lda2) : ap[(int64_t)(k0 + r) * lda2];
    int k = k0;
    for (; k + R <= k1; k += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const d2 av = a[r];
            if (k + R + r < k1) a[r] = NT ? __builtin_nontemporal_load(ap + (int64_t)(k + R + r) * lda2) : ap[(int64_t)(k + R + r) * lda2];
            const double* brow = B + (int64_t)(k + r) * Mp;
#pragma unroll
            for (int j = 0; j < Mp; ++j) {
                const double b = brow[j];
                acc0[j] = __builtin_fma(av.x, b, acc0[j]);
                acc1[j] = __builtin_fma(av.y, b, acc1[j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) if (k + r < k1) {
        const d2 av = a[r];
        const double* brow = B + (int64_t)(k + r) * Mp;
#pragma unroll
        for (int j = 0; j < Mp; ++j) {
            const double b = brow[j];
            acc0[j] = __builtin_fma(av.x, b, acc0[j]);
            acc1[j] = __builtin_fma(av.y, b, acc1[j]);
        }
    }
    // serial reduction of the KW waves through one tile in LDS
    for (int w = 0; w < KW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < Mp; ++j) {
                double* p0 = &red[(2 * lane) * Mp + j];
                double* p1 = &red[(2 * lane + 1) * Mp + j];
                *p0 = (w == 0) ? acc0[j] : *p0 + acc0[j];
                *p1 = (w == 0) ? acc1[j] : *p1 + acc1[j];
            }
        }
        __syncthreads();
    }
    double* dst = out + ((int64_t)blockIdx.y * out_rows + v0) * Mp;
    for (int idx = threadIdx.x; idx < 128 * Mp; idx += 64 * KW) dst[idx] = red[idx];
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; int slots; };

template <int Mp, int KW, int R, bool NT>
Variant mkvf(const double* A, int64_t lda, int64_t K, int64_t vcols, const double* B, double* out, int S) {
    auto kern = vf<Mp, KW, R, NT>;
    size_t lds = (size_t)128 * Mp * 8;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    char buf[200];
    snprintf(buf, 200, "vf Mp=%d KW=%d R=%d NT=%d S=%d blocks=%d bpc=%d", Mp, KW, R, (int)NT, S, (int)(vcols / 128) * S, bpc);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / 128), S), dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)K, S); }, {}, S};
}

Variant mkprod(const double* A, int64_t lda, int64_t K, int64_t vcols, const double* B, double* out, int S) {
    auto kern = gemm_tn_probe_kernel<double, 2, 4, 4, false, 0, 4>;
    size_t lds = (size_t)4 * 64 * 32 * 8;
    char buf[200];
    snprintf(buf, 200, "production gemm_tn (MFMA) S=%d", S);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / 64), S), dim3(256), lds, 0, A, lda, (int64_t)64, B, (const double*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}, S};
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
        printf("%-52s med %7.1f us (min %7.1f max %7.1f) %6.0f GB/s %5.1f TF/s\n", v.name.c_str(), med * 1e3, v.ms.front() * 1e3, v.ms.back() * 1e3,
               gbytes / med * 1e3, tflop / med * 1e3);
    }
    fflush(stdout);
}

void suite(const char* name, int64_t K, int64_t V) {
    const int Mp = 32;
    double *A, *B, *out;
    CK(hipMalloc(&A, 8 * K * V)); CK(hipMalloc(&B, 8 * K * Mp)); CK(hipMalloc(&out, 8 * 40 * V * Mp));
    std::vector<double> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), 8 * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), 8 * K * Mp, hipMemcpyHostToDevice));
    const double gb = 8 * ((double)K * V + 32.0 * (K + V)) / 1e9, tf = 2.0 * K * V * 32 / 1e12;
    printf("== %s: K=%ld V=%ld Mp=32 f64\n", name, (long)K, (long)V);
    std::vector<Variant> vs;
    vs.push_back(mkprod(A, V, K, V, B, out, 6));
    for (int S : {6, 12, 13}) {
        vs.push_back(mkvf<32, 4, 4, false>(A, V, K, V, B, out, S));
        vs.push_back(mkvf<32, 4, 8, false>(A, V, K, V, B, out, S));
        vs.push_back(mkvf<32, 4, 8, true>(A, V, K, V, B, out, S));
    }
    for (int S : {3, 6}) {
        vs.push_back(mkvf<32, 8, 8, false>(A, V, K, V, B, out, S));
        vs.push_back(mkvf<32, 8, 4, false>(A, V, K, V, B, out, S));
    }
    // check first vf variant vs production
    {
        const size_t n1 = (size_t)V * Mp;
        vs[0].launch(); CK(hipDeviceSynchronize());
        std::vector<double> r((size_t)vs[0].slots * n1);
        CK(hipMemcpy(r.data(), out, r.size() * 8, hipMemcpyDeviceToHost));
        for (size_t vi : {1u, 5u, 10u}) {
            vs[vi].launch(); CK(hipDeviceSynchronize());
            std::vector<double> o((size_t)vs[vi].slots * n1);
            CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
            double md = 0, mx = 0;
            for (size_t x = 0; x < n1; ++x) {
                double so = 0, sr = 0;
                for (int s = 0; s < vs[vi].slots; ++s) so += o[s * n1 + x];
                for (int s = 0; s < vs[0].slots; ++s) sr += r[s * n1 + x];
                md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
            }
            printf("check %s: max |diff| %.3e (max |ref| %.3e)\n", vs[vi].name.c_str(), md, mx);
        }
    }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
}

int main() {
    suite("c2_xty", 10048, 5120);     // 5120 = 40 x 128 (padded)
    suite("c2_xw", 5120, 10112);      // 10112 = 79 x 128
    return 0;
}
