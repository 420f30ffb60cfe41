// gemm_probe2.hip - variants of the X-streaming contraction D[v][j] = sum_n A[n][v] B[n][j] (GPU box only).
//   NT   : nontemporal loads of A        ROT : every block starts its K range at a different offset
//   BM   : 0 B from global per wave, 1 B staged through LDS per block (waves tile columns), 2 no B loads (ablation)
//   MODE : 0 full, 1 loads only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
#include "probe_kernels.hpp"
using namespace lcx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int N> struct VecOf;
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<double, 1> { typedef double type; };

// loads RT elements as 16-byte pieces; piece p of lane i sits at column p*16*EPL + i*EPL
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

template <typename T, int CT, int RT, int KW, int U, bool NT, bool ROT, int BM, int MODE>
__global__ void __launch_bounds__(64 * KW)
k2(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, T* __restrict__ out, int64_t out_rows, int K, int nsplit) {
    constexpr int Mp = 16 * CT;
    typedef typename MF<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t v0 = (int64_t)blockIdx.x * (16 * RT);
    const int part = blockIdx.y * KW + wave, nparts = nsplit * KW;
    const int ng = K / (4 * U);
    const int g0 = (int)((int64_t)ng * part / nparts), g1 = (int)((int64_t)ng * (part + 1) / nparts);
    const int cnt = g1 - g0;
    const int rot = ROT ? (int)((blockIdx.x * 2654435761u) % (unsigned)(cnt > 0 ? cnt : 1)) : 0;
    acc_t acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};
    const T* ap = A + v0 + (int64_t)q * lda;
    const T* bp = B + (int64_t)q * Mp + i * CT;
    T a0[U][RT], a1[U][RT];
    Pk<T, CT> b0[U], b1[U];
    if (BM == 2) {
#pragma unroll
        for (int st = 0; st < U; ++st) { b0[st] = ldg<T, CT>(bp + 4 * st * Mp); b1[st] = b0[st]; }
    }
#define LOADG(R, AA, BB) {                                                             \
        int gg = g0 + (R) + rot; if (gg >= g1) gg -= cnt;                                  \
        const int64_t rb = (int64_t)gg * (4 * U);                                          \
        _Pragma("unroll") for (int st = 0; st < U; ++st) {                                 \
            load_a<T, RT, NT>(ap + (rb + 4 * st) * lda, i, AA[st]);                        \
            if (BM == 0) BB[st] = ldg<T, CT>(bp + (rb + 4 * st) * Mp);                     \
        } }
#define MMAG(AA, BB) {                                                                 \
        _Pragma("unroll") for (int st = 0; st < U; ++st)                                   \
        _Pragma("unroll") for (int t = 0; t < RT; ++t)                                     \
        _Pragma("unroll") for (int u = 0; u < CT; ++u) {                                   \
            if (MODE == 1) { asm volatile("" ::"v"(AA[st][t]), "v"(BB[st].v[u])); }        \
            else acc[t][u] = MF<T>::mma(AA[st][t], BB[st].v[u], acc[t][u]);                \
        } }
    if (cnt > 0) {
        LOADG(0, a0, b0);
        int r = 0;
        while (true) {
            int rn = (r + 1 < cnt) ? r + 1 : cnt - 1;
            LOADG(rn, a1, b1);
            MMAG(a0, b0);
            if (++r >= cnt) break;
            rn = (r + 1 < cnt) ? r + 1 : cnt - 1;
            LOADG(rn, a0, b0);
            MMAG(a1, b1);
            if (++r >= cnt) break;
        }
    }
    constexpr int TILE = 16 * RT * Mp;
    constexpr int EPL = 16 / (int)sizeof(T) < RT ? 16 / (int)sizeof(T) : RT;
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
    T* dst = out + ((int64_t)blockIdx.y * out_rows + v0) * Mp;
    for (int idx = threadIdx.x; idx < TILE; idx += 64 * KW) {
        T s = red[idx];
#pragma unroll
        for (int w = 1; w < KW; ++w) s += red[w * TILE + idx];
        dst[idx] = s;
    }
}

template <typename T, int CT, int RT, int KW, int U, bool NT, bool ROT, int BM, int MODE>
double run(const char* tag, const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int S, double gbytes, double tflop) {
    dim3 grid((unsigned)(vcols / (16 * RT)), (unsigned)S);
    size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
    auto kern = k2<T, CT, RT, KW, U, NT, ROT, BM, MODE>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, grid, dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)K, S);
    CK(hipEventRecord(a, 0));
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, grid, dim3(64 * KW), lds, 0, A, lda, B, out, vcols, (int)K, S);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= iters;
    printf("%-8s RT=%d KW=%d U=%d NT=%d ROT=%d BM=%d mode=%d S=%2d blocks=%5u bpc=%d : %7.1f us %6.0f GB/s %5.1f TF/s\n", tag, RT, KW, U, (int)NT, (int)ROT, BM, MODE, S,
           grid.x * grid.y, bpc, ms * 1e3, gbytes / ms * 1e3, tflop / ms * 1e3);
    fflush(stdout);
    return ms;
}


// k3: the KW waves of a block tile adjacent column tiles over the SAME K range; B goes global -> VGPR -> LDS once
// per block and group (double buffered, one barrier per group); the K split is grid.y only.
template <typename T, int CT, int RT, int KW, int U, bool NT, int MODE>
__global__ void __launch_bounds__(64 * KW)
k3(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, T* __restrict__ out, int64_t out_rows, int K, int nsplit, int64_t vcols) {
    constexpr int Mp = 16 * CT;
    constexpr int CHUNK = 4 * U * Mp;                       // elements of B per group
    constexpr int PCS = CHUNK * (int)sizeof(T) / 16;        // 16-byte pieces per group
    constexpr int PPT = (PCS + 64 * KW - 1) / (64 * KW);    // pieces per thread
    typedef typename MF<T>::acc_t acc_t;
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) T Bs[2][CHUNK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t v0 = ((int64_t)blockIdx.x * KW + wave) * (16 * RT);
    const bool active = v0 < vcols;
    const int ng = K / (4 * U);
    const int g0 = (int)((int64_t)ng * blockIdx.y / nsplit), g1 = (int)((int64_t)ng * (blockIdx.y + 1) / nsplit);
    const int cnt = g1 - g0;
    acc_t acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < CT; ++u) acc[t][u] = (acc_t){0, 0, 0, 0};
    const T* ap = A + (active ? v0 : 0) + (int64_t)q * lda;
    T a0[U][RT], a1[U][RT];
    f4 bst[PPT];
#define LOADA(R, AA) {                                                                   \
        const int64_t rb = (int64_t)(g0 + (R)) * (4 * U);                                     \
        _Pragma("unroll") for (int st = 0; st < U; ++st) load_a<T, RT, NT>(ap + (rb + 4 * st) * lda, i, AA[st]); }
#define LOADB(R) {                                                                        \
        const f4* src = reinterpret_cast<const f4*>(B + (int64_t)(g0 + (R)) * CHUNK);          \
        _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                                      \
            const int pc = p * 64 * KW + threadIdx.x;                                          \
            if (PCS % (64 * KW) == 0 || pc < PCS) bst[p] = src[pc]; } }
#define STOREB(BUF) {                                                                     \
        f4* dstp = reinterpret_cast<f4*>(&Bs[BUF][0]);                                         \
        _Pragma("unroll") for (int p = 0; p < PPT; ++p) {                                      \
            const int pc = p * 64 * KW + threadIdx.x;                                          \
            if (PCS % (64 * KW) == 0 || pc < PCS) dstp[pc] = bst[p]; } }
#define MMAL(AA, BUF) {                                                                   \
        Pk<T, CT> bb[U];                                                                       \
        _Pragma("unroll") for (int st = 0; st < U; ++st)                                       \
            bb[st] = *reinterpret_cast<const Pk<T, CT>*>(&Bs[BUF][(4 * st + q) * Mp + i * CT]); \
        _Pragma("unroll") for (int st = 0; st < U; ++st)                                       \
        _Pragma("unroll") for (int t = 0; t < RT; ++t)                                         \
        _Pragma("unroll") for (int u = 0; u < CT; ++u) {                                       \
            if (MODE == 1) { asm volatile("" ::"v"(AA[st][t]), "v"(bb[st].v[u])); }            \
            else acc[t][u] = MF<T>::mma(AA[st][t], bb[st].v[u], acc[t][u]);                    \
        } }
    if (cnt > 0) {
        LOADA(0, a0);
        LOADB(0);
        int r = 0;
        while (true) {
            STOREB(0);
            int rn = (r + 1 < cnt) ? r + 1 : cnt - 1;
            LOADA(rn, a1);
            LOADB(rn);
            __syncthreads();
            MMAL(a0, 0);
            if (++r >= cnt) break;
            STOREB(1);
            rn = (r + 1 < cnt) ? r + 1 : cnt - 1;
            LOADA(rn, a0);
            LOADB(rn);
            __syncthreads();
            MMAL(a1, 1);
            if (++r >= cnt) break;
        }
    }
    if (!active) return;
    constexpr int EPL = 16 / (int)sizeof(T) < RT ? 16 / (int)sizeof(T) : RT;
    T* dst = out + ((int64_t)blockIdx.y * out_rows + v0) * Mp;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = (t / EPL) * 16 * EPL + MF<T>::row(lane, g) * EPL + (t % EPL);
            Pk<T, CT> o;
#pragma unroll
            for (int u = 0; u < CT; ++u) o.v[u] = acc[t][u][g];
            *reinterpret_cast<Pk<T, CT>*>(dst + col * Mp + i * CT) = o;
        }
}

template <typename T, int CT, int RT, int KW, int U, bool NT, int MODE>
double run3(const char* tag, const T* A, int64_t lda, int64_t K, int64_t vcols, const T* B, T* out, int S, double gbytes, double tflop) {
    const int64_t tiles = vcols / (16 * RT);
    dim3 grid((unsigned)((tiles + KW - 1) / KW), (unsigned)S);
    auto kern = k3<T, CT, RT, KW, U, NT, MODE>;
    int bpc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kern, 64 * KW, 0));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, grid, dim3(64 * KW), 0, 0, A, lda, B, out, vcols, (int)K, S, vcols);
    CK(hipEventRecord(a, 0));
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, grid, dim3(64 * KW), 0, 0, A, lda, B, out, vcols, (int)K, S, vcols);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= iters;
    printf("%-8s k3 RT=%d KW=%d U=%d NT=%d mode=%d S=%2d blocks=%5u bpc=%d : %7.1f us %6.0f GB/s %5.1f TF/s\n", tag, RT, KW, U, (int)NT, MODE, S,
           grid.x * grid.y, bpc, ms * 1e3, gbytes / ms * 1e3, tflop / ms * 1e3);
    fflush(stdout);
    return ms;
}

template <typename T, int CT, int RT>
void suite(const char* name, int64_t K, int64_t V, std::initializer_list<int> splits, std::initializer_list<int> splits3, std::initializer_list<int> splits38) {
    T *A, *B, *out, *ref;
    CK(hipMalloc(&A, sizeof(T) * K * V));
    CK(hipMalloc(&B, sizeof(T) * K * 16 * CT));
    CK(hipMalloc(&out, sizeof(T) * 40 * V * 16 * CT));
    CK(hipMalloc(&ref, sizeof(T) * 40 * V * 16 * CT));
    std::vector<T> h((size_t)K * V);
    for (size_t x = 0; x < h.size(); ++x) h[x] = (T)((double)rand() / RAND_MAX - 0.5);
    CK(hipMemcpy(A, h.data(), sizeof(T) * K * V, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), sizeof(T) * K * 16 * CT, hipMemcpyHostToDevice));
    const double gb = sizeof(T) * ((double)K * V + 16.0 * CT * (K + V)) / 1e9, tf = 2.0 * K * V * 16 * CT / 1e12;
    printf("== %s: K=%ld V=%ld Mp=%d elt=%zu RT=%d\n", name, (long)K, (long)V, 16 * CT, sizeof(T), RT);
#define R(KW, U, NT, ROT, BM, MODE, S) run<T, CT, RT, KW, U, NT, ROT, BM, MODE>(name, A, V, K, V, B, out, S, gb, tf)
    for (int S : splits) {
        R(4, 4, false, false, 0, 0, S);
        R(4, 4, true, false, 0, 0, S);
        R(4, 4, true, false, 2, 0, S);
        R(4, 4, true, false, 0, 1, S);
        R(4, 4, true, false, 2, 1, S);
        R(8, 4, true, false, 0, 0, S);
    }
#undef R
#define R3(KW, U, NT, MODE, S) run3<T, CT, RT, KW, U, NT, MODE>(name, A, V, K, V, B, out, S, gb, tf)
    for (int S : splits3) {
        R3(4, 4, true, 0, S);
        R3(4, 4, true, 1, S);
        R3(4, 2, true, 0, S);
        R3(4, 8, true, 0, S);
    }
    for (int S : splits38) { R3(8, 4, true, 0, S); R3(8, 4, false, 0, S); R3(8, 2, true, 0, S); }
#undef R3
    {
        const int S = 5, KW = 4;
        size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
        const int64_t tiles = V / (16 * RT);
        hipLaunchKernelGGL((k3<T, CT, RT, 4, 4, true, 0>), dim3((unsigned)((tiles + KW - 1) / KW), S), dim3(64 * KW), 0, 0, A, V, B, out, V, (int)K, S, V);
        hipLaunchKernelGGL((gemm_tn_probe_kernel<T, CT, RT, 4, false, 0, 4>), dim3((unsigned)(V / (16 * RT)), 3), dim3(64 * KW), lds, 0, A, V, (int64_t)(16 * RT), B, (const T*)nullptr, ref, V, (int)(K / 16), 3, (const int*)nullptr);
        CK(hipDeviceSynchronize());
        const size_t n1 = (size_t)V * 16 * CT;
        std::vector<T> o((size_t)S * n1), r2((size_t)3 * n1);
        CK(hipMemcpy(o.data(), out, o.size() * sizeof(T), hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), ref, r2.size() * sizeof(T), hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s2 = 0; s2 < S; ++s2) so += o[s2 * n1 + x];
            for (int s2 = 0; s2 < 3; ++s2) sr += r2[s2 * n1 + x];
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        printf("k3 check vs production kernel: max |diff| = %.3e (max |ref| = %.3e)\n", md, mx);
    }
    // correctness of the permuted epilogue: NT+ROT result vs the production kernel
    {
        const int S = 3, KW = 4;
        size_t lds = (size_t)KW * 16 * RT * 16 * CT * sizeof(T);
        hipLaunchKernelGGL((k2<T, CT, RT, 4, 4, true, true, 0, 0>), dim3((unsigned)(V / (16 * RT)), S), dim3(64 * KW), lds, 0, A, V, B, out, V, (int)K, S);
        hipLaunchKernelGGL((gemm_tn_probe_kernel<T, CT, RT, 4, false, 0, 4>), dim3((unsigned)(V / (16 * RT)), S), dim3(64 * KW), lds, 0, A, V, (int64_t)(16 * RT), B, (const T*)nullptr, ref, V, (int)(K / 16), S, (const int*)nullptr);
        CK(hipDeviceSynchronize());
        std::vector<T> o((size_t)S * V * 16 * CT), r2(o.size());
        CK(hipMemcpy(o.data(), out, o.size() * sizeof(T), hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), ref, o.size() * sizeof(T), hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        const size_t n1 = (size_t)V * 16 * CT;
        for (size_t x = 0; x < n1; ++x) {
            double so = 0, sr = 0;
            for (int s = 0; s < S; ++s) { so += o[s * n1 + x]; sr += r2[s * n1 + x]; }
            md = fmax(md, fabs(so - sr)); mx = fmax(mx, fabs(sr));
        }
        printf("check vs production kernel: max |diff| = %.3e (max |ref| = %.3e)\n", md, mx);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out)); CK(hipFree(ref));
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "c2";
    if (!strcmp(which, "c2")) {
        suite<double, 2, 4>("c2_xty", 10048, 5056, {6}, {25}, {12});
    } else if (!strcmp(which, "c3")) {
        suite<float, 4, 4>("c3l_xty", 50048, 20032, {1, 2, 3}, {5, 6, 7}, {6, 7, 13});
        suite<float, 4, 4>("c3l_xw", 20032, 50048, {1, 2}, {2, 3, 5}, {2, 3, 5});
    } else if (!strcmp(which, "c4")) {
        suite<float, 8, 2>("c4l_xty", 50048, 20032, {1, 2}, {3, 6, 7}, {6, 7, 13});
        suite<float, 8, 4>("c4l_xty_rt4", 50048, 20032, {1, 2, 3}, {6, 7}, {6, 7, 13});
    }
    return 0;
}
