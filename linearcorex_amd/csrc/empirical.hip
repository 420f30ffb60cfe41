// empirical.hip - gaussianize='empirical' on the device (reference linearcorex.py:424-426):
//     x[:, c]  <-  norm.ppf((rankdata(x[:, c]) - 0.5) / n_samples)        for every column c
// rankdata gives tied values the average of the ranks they span.  The columns are contiguous in the transposed copy of the
// shard (XT[c][0..N)), so the ranks come from one segmented radix sort of (value, row) pairs per chunk of columns - rocPRIM's
// segmented_radix_sort_pairs, a plain library sort on the one-off preprocessing side of the path, not on the fit loop - and
// a kernel of our own turns sorted position into average rank, rank into the normal quantile (Wichura's AS 241 PPND16 in
// double precision, relative error ~1e-16) and scatters it back into both layouts of the shard.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>

#include <rocprim/rocprim.hpp>

namespace lcx {

// normal quantile, AS 241 (PPND16): |relative error| < 1e-16 on (0, 1)
__device__ __forceinline__ double ndtri_as241(double p) {
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r +
                                4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r +
                              1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r +
                                2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r +
                              4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    double r = q < 0.0 ? p : 1.0 - p;
    r = sqrt(-log(r));
    double val;
    if (r <= 5.0) {
        r -= 1.6;
        const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                                1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                              4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                                1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                              2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r -= 5.0;
        const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                                2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                              5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                                7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                              5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return q < 0.0 ? -val : val;
}

struct SegOffset {                       // segment c covers [c * stride + add, ...): add = 0 for the begins, n for the ends
    unsigned int stride, add;
    __host__ __device__ unsigned int operator()(unsigned int c) const { return c * stride + add; }
};

static __global__ void iota_rows_kernel(unsigned int* idx, int64_t ncols, int64_t stride, int64_t n) {
    const int64_t total = ncols * n;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x)
        idx[(k / n) * stride + (k % n)] = (unsigned int)(k % n);
}

// sorted (value, row) pairs of `ncols` columns -> average rank -> quantile, scattered into XT[c0 + c][row] and X[row][c0 + c]
template <typename T>
__global__ void rank_to_quantile_kernel(const T* __restrict__ keys, const unsigned int* __restrict__ rows, int64_t ncols, int64_t stride,
                                        int64_t n, T* __restrict__ X, int64_t ldx, T* __restrict__ XT, int64_t c0) {
    const int64_t total = ncols * n;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = k / n, i = k % n;
        const T* col = keys + c * stride;
        const T v = col[i];
        int64_t lo = i, hi = i + 1;                                   // the tie group [lo, hi) of sorted positions
        if (i > 0 && !(col[i - 1] < v)) {                             // ties below: first position that is not smaller than v
            int64_t a = 0, b = i;
            while (a < b) { const int64_t mid = (a + b) >> 1; if (col[mid] < v) a = mid + 1; else b = mid; }
            lo = a;
        }
        if (i + 1 < n && !(v < col[i + 1])) {                         // ties above: first position that is greater than v
            int64_t a = i + 1, b = n;
            while (a < b) { const int64_t mid = (a + b) >> 1; if (v < col[mid]) b = mid; else a = mid + 1; }
            hi = a;
        }
        const double rank = 0.5 * (double)(lo + 1 + hi);              // average of the 1-based ranks lo+1 .. hi
        const T z = (T)ndtri_as241((rank - 0.5) / (double)n);
        const unsigned int r = rows[c * stride + i];
        XT[(c0 + c) * stride + r] = z;
        X[(int64_t)r * ldx + c0 + c] = z;
    }
}

// X: [Npad][ldx] row-major, XT: [ldx][Npad] its transposed copy (valid on entry); both rewritten for columns [0, V)
template <typename T>
int empirical_columns(T* X, int64_t ldx, T* XT, int64_t Npad, int64_t N, int64_t V, hipStream_t st, std::string* err) {
    auto bad = [&](hipError_t e, const char* what) { *err = std::string(what) + ": " + hipGetErrorString(e); return 1; };
    int64_t chunk = ((int64_t)1 << 27) / Npad;                        // <= 128M elements per sort
    if (chunk < 1) chunk = 1;
    if (chunk > V) chunk = V;
    if ((double)chunk * (double)Npad >= 4.0e9) { *err = "empirical: n_samples too large for one sort segment"; return 1; }
    T* keys_out = nullptr;
    unsigned int *rows_in = nullptr, *rows_out = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e;
    const size_t elems = (size_t)chunk * Npad;
    if ((e = hipMalloc((void**)&keys_out, elems * sizeof(T))) != hipSuccess) return bad(e, "empirical: hipMalloc");
    if ((e = hipMalloc((void**)&rows_in, elems * 4)) != hipSuccess) { (void)hipFree(keys_out); return bad(e, "empirical: hipMalloc"); }
    if ((e = hipMalloc((void**)&rows_out, elems * 4)) != hipSuccess) { (void)hipFree(keys_out); (void)hipFree(rows_in); return bad(e, "empirical: hipMalloc"); }
    int rc = 0;
    hipLaunchKernelGGL(iota_rows_kernel, dim3(2048), dim3(256), 0, st, rows_in, chunk, Npad, N);
    typedef rocprim::counting_iterator<unsigned int> Cnt;
    typedef rocprim::transform_iterator<Cnt, SegOffset, unsigned int> It;
    for (int64_t c0 = 0; c0 < V && rc == 0; c0 += chunk) {
        const int64_t nc = (V - c0) < chunk ? (V - c0) : chunk;
        const T* keys_in = XT + c0 * Npad;
        It b(Cnt(0), SegOffset{(unsigned int)Npad, 0u});
        It en(Cnt(0), SegOffset{(unsigned int)Npad, (unsigned int)N});
        size_t need = 0;
        e = rocprim::segmented_radix_sort_pairs(nullptr, need, keys_in, keys_out, rows_in, rows_out, (unsigned int)(nc * Npad),
                                                (unsigned int)nc, b, en, 0, 8 * sizeof(T), st);
        if (e != hipSuccess) { rc = bad(e, "empirical: sort sizing"); break; }
        if (need > temp_bytes) {
            if (temp) (void)hipFree(temp);
            temp = nullptr;
            if ((e = hipMalloc(&temp, need)) != hipSuccess) { rc = bad(e, "empirical: hipMalloc of the sort's temporary storage"); break; }
            temp_bytes = need;
        }
        e = rocprim::segmented_radix_sort_pairs(temp, need, keys_in, keys_out, rows_in, rows_out, (unsigned int)(nc * Npad),
                                                (unsigned int)nc, b, en, 0, 8 * sizeof(T), st);
        if (e != hipSuccess) { rc = bad(e, "empirical: segmented_radix_sort_pairs"); break; }
        hipLaunchKernelGGL((rank_to_quantile_kernel<T>), dim3(4096), dim3(256), 0, st, (const T*)keys_out, (const unsigned int*)rows_out, nc,
                           Npad, N, X, ldx, XT, c0);
        if ((e = hipGetLastError()) != hipSuccess) { rc = bad(e, "empirical: rank kernel"); break; }
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess && rc == 0) rc = bad(e, "empirical: synchronize");
    (void)hipFree(keys_out); (void)hipFree(rows_in); (void)hipFree(rows_out);
    if (temp) (void)hipFree(temp);
    return rc;
}

template int empirical_columns<float>(float*, int64_t, float*, int64_t, int64_t, int64_t, hipStream_t, std::string*);
template int empirical_columns<double>(double*, int64_t, double*, int64_t, int64_t, int64_t, hipStream_t, std::string*);

}  // namespace lcx
