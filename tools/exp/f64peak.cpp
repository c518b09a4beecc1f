// f64peak.cpp — what the vector pipe sustains for dependent-free v_fma_f64 streams (register operands, and with one
// operand from an SGPR pair as pitch_strength_kernel / iir_zero_state_kernel use it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool SCALAR>
__global__ void __launch_bounds__(256) fma_kernel(double* out, const double* __restrict__ coef, int iters) {
    double acc[32];
    const double x = (double)threadIdx.x * 1e-9 + 1.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (double)i;
    for (int it = 0; it < iters; ++it) {
        if (SCALAR) {
            const double c = coef[it & 63];                   // wave-uniform -> SGPR pair
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(c, x, acc[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(acc[i], x, x);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;
    double *out, *coef;
    HK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    HK(hipMalloc(&coef, 64 * 8));
    HK(hipMemset(coef, 0, 64 * 8));
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            HK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(fma_kernel<false>, dim3(blocks), dim3(256), 0, 0, out, coef, iters);
            else hipLaunchKernelGGL(fma_kernel<true>, dim3(blocks), dim3(256), 0, 0, out, coef, iters);
            HK(hipEventRecord(e1, 0));
            HK(hipEventSynchronize(e1));
            float ms;
            HK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 2.0 * 32 * iters * (double)blocks * 256;
            printf("%s operands: %.2f ms  %.1f TFLOP/s f64\n", mode ? "SGPR x VGPR" : "VGPR", ms, flops / ms * 1e-9);
        }
    }
    return 0;
}
