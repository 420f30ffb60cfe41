// gemm_probe3.hip - ring-pipelined, stream-K balanced variant of the X-streaming contraction (GPU box only).
//   D[v][j] = sum_n A[n][v] B[n][j];  unit of work = (column tile, step of 4 rows); every block gets the same
//   number of units (+-1), a block's KW waves split each tile segment, partials go to slot (block - first block
//   of the tile).  Per wave a ring of NB one-step register buffers keeps NB-1 steps of loads in flight.
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

template <typename T, int N> struct VecOf;
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };

template <typename T, int RT, bool NT>
__device__ __forceinline__ void load_a(const T* rowp, int i, T (&dst)[RT]) {
    constexpr int EPL = 16 / (int)sizeof(T) < RT ? 16 / (int)sizeof(T) : RT;
    typedef typename VecOf<T, EPL>::type V;
#pragma unroll
    for (int p = 0; p < RT / EPL; ++p) {
        const V* src = reinterpret_cast<const V*>(rowp + p * 16 * EPL + i * EPL);
        V v = NT ? __builtin_nontemporal_load(src) : *src;
#pragma unroll
        for (int e = 0; e < EPL; ++e) dst[p * EPL + e] = v[e];
    }
}

__device__ __forceinline__ int sk_owner(int64_t L, int64_t total, int nb) { return (int)(((L + 1) * nb - 1) / total); }

template <typename T, int CT, int RT, int KW, int NB, bool NT, int MODE>
__global__ void __launch_bounds__(64 * KW)
k4(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, T* __restrict__ out, int64_t out_rows, int K, int ntiles, int maxslots, long long* clk) {
    long long c0 = 0, r0 = 0;
    if (clk && threadIdx.x == 0 && blockIdx.x == 7) { c0 = clock64(); r0 = wall_clock64(); }
    constexpr int Mp = 16 * CT;
    typedef typename MF<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int ng = K / 4;
    const int64_t total = (int64_t)ntiles * ng;
    const int nb = gridDim.x;
    int64_t L0 = total * blockIdx.x / nb;
    const int64_t L1 = total * (blockIdx.x + 1) / nb;
    constexpr int TILE = 16 * RT * Mp;
    constexpr int EPL = 16 / (int)sizeof(T) < RT ? 16 / (int)sizeof(T) : RT;
    while (L0 < L1) {
        const int tile = (int)(L0 / ng);
        const int s0 = (int)(L0 - (int64_t)tile * ng);
        const int s1 = (int)((L1 - L0) < (int64_t)(ng - s0) ? s0 + (L1 - L0) : ng);
        const int len = s1 - s0;
        const int w0 = s0 + (int)((int64_t)len * wave / KW), w1 = s0 + (int)((int64_t)len * (wave + 1) / KW);
        acc_t acc[RT][CT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};
        const T* ap = A + (int64_t)tile * (16 * RT) + (int64_t)q * lda;
        const T* bp = B + (int64_t)q * Mp + i * CT;
        T a[NB][RT];
        Pk<T, CT> b[NB];
#define LD(J, S) { load_a<T, RT, NT>(ap + (int64_t)(S) * 4 * lda, i, a[J]); b[J] = ldg<T, CT>(bp + (int64_t)(S) * 4 * Mp); }
#define MM(J) { _Pragma("unroll") for (int t = 0; t < RT; ++t) _Pragma("unroll") for (int u = 0; u < CT; ++u) {   \
                if (MODE == 1) { asm volatile("" ::"v"(a[J][t]), "v"(b[J].v[u])); }                                  \
                else acc[t][u] = MF<T>::mma(a[J][t], b[J].v[u], acc[t][u]); } }
#pragma unroll
        for (int j = 0; j < NB; ++j) if (w0 + j < w1) LD(j, w0 + j);
        int s = w0;
        for (; s + NB <= w1; s += NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                MM(j);
                if (MODE != 2 && s + NB + j < w1) LD(j, s + NB + j);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) if (s + j < w1) MM(j);
#undef LD
#undef MM
        T* mine = red + wave * TILE;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int u = 0; u < CT; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = (t / EPL) * 16 * EPL + MF<T>::row(lane, g) * EPL + (t % EPL);
                    mine[col * Mp + i * CT + u] = acc[t][u][g];
                }
        __syncthreads();
        const int fb = sk_owner((int64_t)tile * ng, total, nb);
        const int slot = blockIdx.x - fb;
        T* dst = out + ((int64_t)slot * out_rows + (int64_t)tile * 16 * RT) * Mp;
        for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) {
            T sacc = red[idx];
#pragma unroll
            for (int w = 1; w < KW; ++w) sacc += red[w * TILE + idx];
            dst[idx] = sacc;
        }
        if (s1 == ng) {      // last contributor of this tile: zero the unused slots
            const int lb = sk_owner((int64_t)tile * ng + ng - 1, total, nb);
            for (int sl = lb - fb + 1; sl < maxslots; ++sl) {
                T* z = out + ((int64_t)sl * out_rows + (int64_t)tile * 16 * RT) * Mp;
                for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) z[idx] = (T)0;
            }
        }
        __syncthreads();
        L0 += len;
    }
    if (clk && threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

static long long* g_clk = nullptr;
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

template <typename T, int CT, int RT, int KW, int NB, bool NT, int MODE>
Variant mk4(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int bpc_want) {
    auto kern = k4<T, CT, RT, KW, NB, NT, MODE>;
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    const int use = bpc_want > 0 && bpc_want < bpc ? bpc_want : bpc;
    const int nb = 256 * use;
    const int ntiles = (int)(vcols / (16 * RT));
    const int maxslots = (nb + ntiles - 1) / ntiles + 1;
    char buf[200];
    snprintf(buf, 200, "k4 RT=%d KW=%d NB=%2d NT=%d mode=%d bpc=%d(use %d) nb=%d slots=%d", RT, KW, NB, (int)NT, MODE, bpc, use, nb, maxslots);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)K, ntiles, maxslots, g_clk); }, {}};
}


template <typename T, int CT, int RT, int KW, int U, int MINW, int PRIO>
__global__ void __launch_bounds__(64 * KW, MINW)
k5(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, T* __restrict__ out, int64_t out_rows, int K, int nsplit) {
    constexpr int Mp = 16 * CT;
    typedef typename MF<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);      // ONE tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t v0 = (int64_t)blockIdx.x * (16 * RT);
    const int part = blockIdx.y * KW + wave, nparts = nsplit * KW;
    const int ng = K / (4 * U);
    const int g0 = (int)((int64_t)ng * part / nparts), g1 = (int)((int64_t)ng * (part + 1) / nparts);
    acc_t acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};
    const T* ap = A + v0 + (int64_t)q * lda;
    const T* bp = B + (int64_t)q * Mp + i * CT;
    T a0[U][RT], a1[U][RT];
    Pk<T, CT> b0[U], b1[U];
#define LOADG(G, AA, BB) {                                                             \
        const int64_t rb = (int64_t)(G) * (4 * U);                                         \
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);                                      \
        _Pragma("unroll") for (int st = 0; st < U; ++st) {                                 \
            load_a<T, RT, false>(ap + (rb + 4 * st) * lda, i, AA[st]);                     \
            BB[st] = ldg<T, CT>(bp + (rb + 4 * st) * Mp);                                  \
        }                                                                                  \
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0); }
#define MMAG(AA, BB) {                                                                 \
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                      \
        _Pragma("unroll") for (int st = 0; st < U; ++st)                                   \
        _Pragma("unroll") for (int t = 0; t < RT; ++t)                                     \
        _Pragma("unroll") for (int u = 0; u < CT; ++u)                                     \
            acc[t][u] = MF<T>::mma(AA[st][t], BB[st].v[u], acc[t][u]);                     \
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0); }
    if (g0 < g1) {
        LOADG(g0, a0, b0);
        int g = g0;
        while (true) {
            int gn = (g + 1 < g1) ? g + 1 : g1 - 1;
            LOADG(gn, a1, b1);
            MMAG(a0, b0);
            if (++g >= g1) break;
            gn = (g + 1 < g1) ? g + 1 : g1 - 1;
            LOADG(gn, a0, b0);
            MMAG(a1, b1);
            if (++g >= g1) break;
        }
    }
#undef LOADG
#undef MMAG
    constexpr int TILE = 16 * RT * Mp;
    constexpr int EPL = 16 / (int)sizeof(T) < RT ? 16 / (int)sizeof(T) : RT;
    for (int w = 0; w < KW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int u = 0; u < CT; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = (t / EPL) * 16 * EPL + MF<T>::row(lane, g) * EPL + (t % EPL);
                        T* p = &red[col * Mp + i * CT + u];
                        *p = (w == 0) ? acc[t][u][g] : *p + acc[t][u][g];
                    }
        }
        __syncthreads();
    }
    T* dst = out + ((int64_t)blockIdx.y * out_rows + v0) * Mp;
    for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) dst[idx] = red[idx];
}

template <typename T, int CT, int RT, int KW, int U, int MINW, int PRIO>
Variant mk5(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int S) {
    auto kern = k5<T, CT, RT, KW, U, MINW, PRIO>;
    size_t lds = (size_t)16 * RT * 16 * CT * sizeof(T);
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    char buf[200];
    snprintf(buf, 200, "k5 RT=%d KW=%d U=%d minw=%d prio=%d S=%d blocks=%d bpc=%d", RT, KW, U, MINW, PRIO, S, (int)(vcols / (16 * RT)) * S, bpc);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)K, S); }, {}};
}

template <typename T, int CT, int RT, int KW>
Variant mkprod(const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int S) {
    auto kern = gemm_tn_probe_kernel<T, CT, RT, KW, false, 0, 4>;
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    char buf[200];
    snprintf(buf, 200, "production gemm_tn RT=%d KW=%d S=%d blocks=%d", RT, KW, S, (int)(vcols / (16 * RT)) * S);
    return Variant{buf, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)(vcols / (16 * RT)), S), dim3(64 * KW), lds, 0, A, lda, (int64_t)(16 * RT), B, (const T*)nullptr, out, vcols, (int)(K / 16), S, (const int*)nullptr); }, {}};
}

static void bench(std::vector<Variant>& vs, double gbytes, double tflop, int rounds = 7, int iters = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& v : vs) { v.launch(); }
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
        long long hc[2] = {0, 0};
        CK(hipMemset(g_clk, 0, 16));
        if (v.name[1] == '4') { v.launch(); v.launch(); }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc, g_clk, 16, hipMemcpyDeviceToHost));
        printf("%-64s med %6.1f us (min %6.1f max %6.1f) %6.0f GB/s %5.1f TF/s  shader clk %4.0f MHz\n", v.name.c_str(), med * 1e3, v.ms.front() * 1e3, v.ms.back() * 1e3,
               gbytes / med * 1e3, tflop / med * 1e3, hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0);
    }
    fflush(stdout);
}

template <typename T, int CT, int RT>
bool check(const T* A, int64_t K, int64_t V, const T* B, T* out, T* ref, Variant& v, int maxslots) {
    CK(hipMemset(out, 0xff, sizeof(T) * maxslots * V * 16 * CT));
    v.launch();
    const int KW = 4;
    size_t lds = (size_t)KW * 16 * 4 * 16 * CT * sizeof(T);
    hipLaunchKernelGGL((gemm_tn_probe_kernel<T, CT, 4, 4, false, 0, 4>), dim3((unsigned)(V / 64), 3), dim3(64 * KW), lds, 0, A, V, (int64_t)64, B, (const T*)nullptr, ref, V, (int)(K / 16), 3, (const int*)nullptr);
    CK(hipDeviceSynchronize());
    const size_t n1 = (size_t)V * 16 * CT;
    std::vector<T> o((size_t)maxslots * n1), r2((size_t)3 * n1);
    CK(hipMemcpy(o.data(), out, o.size() * sizeof(T), hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), ref, r2.size() * sizeof(T), hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t x = 0; x < n1; ++x) {
        double so = 0, sr = 0;
        for (int s2 = 0; s2 < maxslots; ++s2) so += o[s2 * n1 + x];
        for (int s2 = 0; s2 < 3; ++s2) sr += r2[s2 * n1 + x];
        md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
    }
    printf("check %s: max |diff| = %.3e (max |ref| = %.3e)\n", v.name.c_str(), md, mx);
    return md < 1e-6 * mx;
}

template <typename T, int CT>
void suite(const char* name, int64_t K, int64_t V, bool zero = false) {
    T *A, *B, *out, *ref;
    CK(hipMalloc(&A, sizeof(T) * K * V));
    CK(hipMalloc(&B, sizeof(T) * K * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 40 * V * 16 * CT));
    CK(hipMalloc(&ref, sizeof(T) * 8 * V * 16 * CT));
    std::vector<T> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = zero ? (T)0 : (T)((double)rand() / RAND_MAX - 0.5);
    if (!g_clk) CK(hipMalloc(&g_clk, 64));
    CK(hipMemcpy(A, h.data(), sizeof(T) * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), sizeof(T) * K * 16 * CT, hipMemcpyHostToDevice));
    const double gb = sizeof(T) * ((double)K * V + 16.0 * CT * (K + V)) / 1e9, tf = 2.0 * K * V * 16 * CT / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d elt=%zu\n", name, (long)K, (long)V, 16 * CT, sizeof(T));
    std::vector<Variant> vs;
    vs.push_back(mkprod<T, CT, 4, 4>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 4, 4, 1, 0>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 4, 4, 1, 1>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 4, 4, 1, 2>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 4, 4, 3, 0>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 4, 4, 3, 0>(A, V, K, V, B, out, 9));
    vs.push_back(mk5<T, CT, 4, 4, 2, 4, 0>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 4, 2, 4, 0>(A, V, K, V, B, out, 12));
    vs.push_back(mk5<T, CT, 4, 4, 2, 4, 0>(A, V, K, V, B, out, 13));
    vs.push_back(mk5<T, CT, 4, 4, 2, 3, 0>(A, V, K, V, B, out, 9));
    vs.push_back(mk5<T, CT, 4, 2, 2, 4, 0>(A, V, K, V, B, out, 12));
    vs.push_back(mk5<T, CT, 4, 2, 2, 4, 0>(A, V, K, V, B, out, 26));
    vs.push_back(mk5<T, CT, 4, 8, 2, 4, 0>(A, V, K, V, B, out, 3));
    vs.push_back(mk5<T, CT, 4, 8, 2, 4, 0>(A, V, K, V, B, out, 6));
    vs.push_back(mk5<T, CT, 4, 8, 4, 2, 0>(A, V, K, V, B, out, 3));
    if (!zero) { g_clk = nullptr; check<T, CT, 4>(A, K, V, B, out, ref, vs[1], 6); check<T, CT, 4>(A, K, V, B, out, ref, vs[6], 6); CK(hipMalloc(&g_clk, 64)); }
    bench(vs, gb, tf);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out)); CK(hipFree(ref));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "c2";
    if (!strcmp(which, "c2")) {
        suite<double, 2>("c2_xty", 10048, 5056);
        suite<double, 2>("c2_xw", 5056, 10048);
    } else if (!strcmp(which, "c3")) {
        suite<float, 4>("c3l_xty", 50048, 20032);
        suite<float, 4>("c3l_xw", 20032, 50048);
    }
    return 0;
}
