// mfma_map.hip - empirical lane layout of v_mfma_f64_4x4x4f64 (4 blocks) on gfx950 (GPU box only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void probe(double* out) {          // out[la][lane] = D when A = onehot(la), B[l] = 100 + l
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        double a = lane == la ? 1.0 : 0.0, b = 100.0 + lane;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + lane] = d;
    }
}
int main() {
    double* o; CK(hipMalloc(&o, 64 * 64 * 8));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o);
    CK(hipDeviceSynchronize());
    static double h[64 * 64];
    CK(hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost));
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d ->", la);
        for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0.0) printf(" D[%2d]=B[%2d]", l, (int)(h[la * 64 + l] - 100.0));
        printf("\n");
    }
    return 0;
}
