// gemm_probe9.hip - the float32 contractions of the large shards on the bf16 matrix pipe (3-way split, gemm_split_kernels.hpp)
// against the production float32-MFMA kernels on the same panel-major copy: time, and error against a float64 contraction
// (GPU box only).   usage: gemm_probe9 [c3|c4shard|all]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <functional>
#include <string>
#include <string.h>
#include <type_traits>
#include "probe_kernels.hpp"
#include "../linearcorex_amd/csrc/gemm_split_kernels.hpp"
using namespace lcx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; int maxslots; };

static void geometry(const void* kern, int threads, int bpc_use, int64_t rows, int rows_per_block, int ng, int* nb, int* nsuper, int* maxslots, int* bpc) {
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(bpc, kern, threads, 0));
    const int use = bpc_use > 0 ? bpc_use : *bpc;
    *nsuper = (int)((rows + rows_per_block - 1) / rows_per_block);
    const int64_t total = (int64_t)*nsuper * ng;
    *nb = 256 * use;
    if (*nb > total) *nb = (int)total;
    *maxslots = (*nb + *nsuper - 1) / *nsuper + 1;
}

// production float32 kernels on the panel-major copy
template <int CT, int KW, bool CONTRACT_N>
Variant mkprod(const float* XP, int64_t nrows_pad, int64_t K, int64_t rows, const float* B, float* out, int bpc_use) {
    constexpr int RT = CtShape<float, CT>::RT, U = 4;
    const void* kern = CONTRACT_N ? (const void*)gemm_ct_kernel<float, CT, RT, KW, U, true, true> : (const void*)gemm_cr_kernel<float, CT, RT, KW, U, true, true>;
    int nb, nsuper, maxslots, bpc;
    const int ng = (int)(K / (4 * U));
    geometry(kern, 64 * KW, bpc_use, rows, KW * 16 * RT, ng, &nb, &nsuper, &maxslots, &bpc);
    char buf[200];
    snprintf(buf, 200, "f32 MFMA %s KW=%d bpc=%d(use %d) slots=%d", CONTRACT_N ? "gemm_ct" : "gemm_cr", KW, bpc, bpc_use, maxslots);
    const int64_t ps = nrows_pad * 16;
    if (CONTRACT_N)
        return Variant{buf, [=] { hipLaunchKernelGGL((gemm_ct_kernel<float, CT, RT, KW, U, true, true>), dim3(nb), dim3(64 * KW), 0, 0, XP, ps, B, out, rows, rows, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
    return Variant{buf, [=] { hipLaunchKernelGGL((gemm_cr_kernel<float, CT, RT, KW, U, true, true>), dim3(nb), dim3(64 * KW), 0, 0, XP, ps, B, out, rows, rows, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

template <int CT, int KW, int NP, bool CONTRACT_N, bool NT, bool PF, int WPE, int KS = 1, int PRIO = 0>
Variant mksplit(const float* XP, int64_t nrows_pad, int64_t K, int64_t rows, const float* B, u32x4_t* Bsp, float* out, int bpc_use) {
    auto kern = gemm_split_kernel<CT, KW, NP, CONTRACT_N, NT, PF, WPE, KS, PRIO>;
    int nb, nsuper, maxslots, bpc;
    const int ng = (int)(K / (SPLIT_KG * KS));
    geometry((const void*)kern, 64 * KW, bpc_use, rows, KW * 64, ng, &nb, &nsuper, &maxslots, &bpc);
    char buf[200];
    snprintf(buf, 200, "bf16 x %d KW=%d pf=%d wpe=%d ks=%d prio=%d nt=%d bpc=%d(use %d) slots=%d", NP, KW, (int)PF, WPE, KS, PRIO, (int)NT, bpc, bpc_use, maxslots);
    const int64_t ps = nrows_pad * 16;
    return Variant{buf, [=] {
        hipLaunchKernelGGL((split_b_kernel<CT, CONTRACT_N>), dim3(1024), dim3(256), 0, 0, B, Bsp, ng * KS, (const int*)nullptr);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), 0, 0, XP, ps, (const u32x4_t*)Bsp, out, rows, rows, ng, nsuper, maxslots, (const int*)nullptr); }, {}, maxslots};
}

static void bench(std::vector<Variant>& vs, double gbytes, double tflop, int rounds = 5, int iters = 5) {
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
        printf("%-56s med %8.1f us (min %8.1f max %8.1f) %6.0f GB/s %6.1f TF/s\n", v.name.c_str(), med * 1e3, v.ms.front() * 1e3, v.ms.back() * 1e3,
               gbytes / med * 1e3, tflop / med * 1e3);
    }
    fflush(stdout);
}

// float64 contraction of 64 sample output rows: ref[r][j]
__global__ void ref64_kernel(const float* X, int64_t V, int64_t N, const float* B, int Mp, int contract_n, int64_t row0, int64_t row_step, double* ref) {
    const int r = blockIdx.x, j = threadIdx.x;
    const int64_t row = row0 + r * row_step;
    double s = 0;
    if (contract_n) for (int64_t n = 0; n < N; ++n) s += (double)X[n * V + row] * (double)B[n * Mp + j];
    else            for (int64_t v = 0; v < V; ++v) s += (double)X[row * V + v] * (double)B[v * Mp + j];
    ref[r * Mp + j] = s;
}

static void accuracy(Variant& v, float* out, int64_t rows, int Mp, const std::vector<double>& ref, int64_t row0, int64_t row_step, int nref) {
    const size_t n1 = (size_t)rows * Mp;
    CK(hipMemset(out, 0xff, sizeof(float) * v.maxslots * n1));
    v.launch();
    CK(hipDeviceSynchronize());
    std::vector<float> o((size_t)v.maxslots * n1);
    CK(hipMemcpy(o.data(), out, o.size() * sizeof(float), hipMemcpyDeviceToHost));
    double mx = 0, se = 0, sr = 0;
    for (int r = 0; r < nref; ++r)
        for (int j = 0; j < Mp; ++j) {
            double so = 0;
            for (int s2 = 0; s2 < v.maxslots; ++s2) so += o[s2 * n1 + (size_t)(row0 + r * row_step) * Mp + j];
            const double d = so - ref[(size_t)r * Mp + j];
            mx = fmax(mx, fabs(d)); se += d * d; sr += ref[(size_t)r * Mp + j] * ref[(size_t)r * Mp + j];
        }
    const double n = (double)nref * Mp;
    printf("error vs f64  %-56s max %.3e  rms %.3e  (rms |ref| %.3e; rms err / rms ref %.2e)\n", v.name.c_str(), mx, sqrt(se / n), sqrt(sr / n), sqrt(se / sr));
}

template <int CT>
void suite(const char* name, int64_t N, int64_t V) {
    constexpr int Mp = 16 * CT, NREF = 64;
    float *X, *XP, *B, *out;
    u32x4_t* Bsp;
    double* ref;
    CK(hipMalloc(&X, sizeof(float) * N * V));
    CK(hipMalloc(&XP, sizeof(float) * N * V));
    const int64_t big = std::max(N, V);
    CK(hipMalloc(&B, sizeof(float) * big * Mp));
    CK(hipMalloc(&out, sizeof(float) * 40 * big * Mp));
    CK(hipMalloc(&ref, sizeof(double) * NREF * Mp));
    CK(hipMalloc(&Bsp, (size_t)6 * big * Mp));
    {
        // Gaussian-ish entries (sum of 4 uniforms, unit variance), asymmetric B
        std::vector<float> h((size_t)4096 * 4099);
        for (size_t x = 0; x < h.size(); ++x) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += (double)rand() / RAND_MAX - 0.5;
            h[x] = (float)(s * sqrt(3.0));
        }
        for (size_t off = 0; off < (size_t)N * V; off += h.size())
            CK(hipMemcpy(X + off, h.data(), sizeof(float) * std::min(h.size(), (size_t)N * V - off), hipMemcpyHostToDevice));
        CK(hipMemcpy(B, h.data() + 11, sizeof(float) * big * Mp, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL((panelize_kernel<float>), dim3(4096), dim3(256), 0, 0, X, V, XP, N, V);
    CK(hipDeviceSynchronize());
    const double gb = 4.0 * ((double)N * V + (double)Mp * (N + V)) / 1e9, tf = 2.0 * N * V * Mp / 1e12;
    std::vector<double> href((size_t)NREF * Mp);
    for (int pass = 0; pass < 2; ++pass) {
        const bool cn = pass == 1;
        const int64_t rows = cn ? V : N, K = cn ? N : V;
        const int64_t row0 = 5, row_step = rows / NREF - 1;
        hipLaunchKernelGGL(ref64_kernel, dim3(NREF), dim3(Mp), 0, 0, X, V, N, B, Mp, (int)cn, row0, row_step, ref);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(href.data(), ref, sizeof(double) * NREF * Mp, hipMemcpyDeviceToHost));
        printf("== %s: X %ld x %ld float32, Mp=%d: %s from the panel-major copy\n", name, (long)N, (long)V, Mp, cn ? "X^T.Y (contraction over samples)" : "X.B^T (contraction over variables)");
        std::vector<Variant> vs;
        auto add = [&](auto cn_tag) {
            constexpr bool CN = decltype(cn_tag)::value;
            vs.push_back(mkprod<CT, CtShape<float, CT>::KW, CN>(XP, N, K, rows, B, out, CT == 8 ? 1 : 2));
            if constexpr (CT <= 4) {
                vs.push_back(mksplit<CT, 4, 6, CN, true, false, 2, 2, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 4, 6, CN, true, false, 2, 2, 2>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 6, CN, true, false, 2, 2, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 6, CN, true, false, 2, 2, 2>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 6, CN, true, true, 2, 2, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 3, CN, true, false, 2, 2, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 8, CN, true, false, 2, 2, 1>(XP, N, K, rows, B, Bsp, out, 0));
            } else {
                vs.push_back(mksplit<CT, 8, 6, CN, true, false, 2, 1, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 6, CN, true, false, 2, 1, 2>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 6, CN, true, false, 2, 1, 0>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 4, 6, CN, true, false, 1, 1, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 3, CN, true, false, 2, 1, 1>(XP, N, K, rows, B, Bsp, out, 0));
                vs.push_back(mksplit<CT, 8, 8, CN, true, false, 2, 1, 1>(XP, N, K, rows, B, Bsp, out, 0));
            }
        };
        if (cn) add(std::true_type{}); else add(std::false_type{});
        for (auto& v : vs) accuracy(v, out, rows, Mp, href, row0, row_step, NREF);
        bench(vs, gb, tf);
    }
    CK(hipFree(X)); CK(hipFree(XP)); CK(hipFree(B)); CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(Bsp));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(which, "all");
    if (all || !strcmp(which, "small")) suite<4>("small", 2048, 4096);
    if (all || !strcmp(which, "c3")) suite<4>("c3", 50048, 100032);
    if (all || !strcmp(which, "c4shard")) suite<8>("c4shard", 50048, 125056);
    return 0;
}
