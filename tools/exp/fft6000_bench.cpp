// fft6000_bench.cpp — the 6000-point complex float64 workgroup transform: run-time mixed radix (fft_mixed.h) against the
// compile-time plan 6 x 10 x 10 x 10 (fft_static.h): agreement and time per transform, one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifriture_amd/csrc -Iinclude -o tools/exp/fft6000_bench.bin tools/exp/fft6000_bench.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "fft_static.h"
using namespace frt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 6000, NT = 1024;

template <int MODE>
__global__ void __launch_bounds__(NT) k(const double* in, double* out, const double* tw, MixedPlan plan, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx<double>* buf = (cpx<double>*)smem;
    const int tid = threadIdx.x;
    cpx<double>* twl = buf + N;                        // MODE 2: the plan's tables copied to LDS
    if (MODE == 2)
        for (int i = tid; i < 666; i += NT) twl[i] = ((const cpx<double>*)tw)[i];
    for (int i = tid; i < N; i += NT) buf[i] = {in[2 * i], in[2 * i + 1]};
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) fft_mixed_forward<double, 3>(buf, (const cpx<double>*)tw, plan, tid, NT);
        else if (MODE == 1) static_fft_forward<double, NT, 6, 10, 10, 10>(buf, (const cpx<double>*)tw, tid);
        else static_fft_forward<double, NT, 6, 10, 10, 10>(buf, twl, tid);
        if (it + 1 < iters) {
            for (int i = tid; i < N; i += NT) buf[i] = {buf[i].x * (1.0 / 77.0), buf[i].y * (1.0 / 77.0)};
            __syncthreads();
        }
    }
    if (blockIdx.x == 0)
        for (int i = tid; i < N; i += NT) { out[2 * i] = buf[i].x; out[2 * i + 1] = buf[i].y; }
}

int main() {
    std::vector<double> h(2 * N);
    for (int i = 0; i < 2 * N; ++i) h[i] = sin(0.37 * i) + 0.25 * cos(1.7 * i * i * 1e-3);
    MixedPlan plan;
    make_mixed_plan(N, &plan);
    auto twm = make_pass_twiddles<double>(plan);
    auto tws = make_static_twiddles<double>({6, 10, 10, 10});
    double *din, *dout, *dtwm, *dtws;
    CK(hipMalloc(&din, h.size() * 8)); CK(hipMalloc(&dout, h.size() * 8));
    CK(hipMalloc(&dtwm, twm.size() * 8)); CK(hipMalloc(&dtws, tws.size() * 8));
    CK(hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwm, twm.data(), twm.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtws, tws.data(), tws.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(N + 672) * 16;
    CK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<double> r0(2 * N), r1(2 * N);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(NT), lds, 0, din, dout, dtwm, plan, 1);
    CK(hipMemcpy(r0.data(), dout, r0.size() * 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(NT), lds, 0, din, dout, dtws, plan, 1);
    CK(hipMemcpy(r1.data(), dout, r1.size() * 8, hipMemcpyDeviceToHost));
    // direct DFT of a few bins in long double
    double maxref = 0, e01 = 0, eref = 0;
    for (int i = 0; i < 2 * N; ++i) { maxref = fmax(maxref, fabs(r0[i])); e01 = fmax(e01, fabs(r0[i] - r1[i])); }
    for (int kk : {0, 1, 7, 599, 600, 3000, 5999}) {
        long double re = 0, im = 0;
        for (int n = 0; n < N; ++n) {
            const long double a = -6.283185307179586476925286766559L * (long double)((long long)n * kk % N) / N;
            re += h[2 * n] * cosl(a) - h[2 * n + 1] * sinl(a);
            im += h[2 * n] * sinl(a) + h[2 * n + 1] * cosl(a);
        }
        eref = fmax(eref, fmax(fabs((double)re - r1[2 * kk]), fabs((double)im - r1[2 * kk + 1])));
    }
    printf("max |X| %.3e   mixed vs static %.3e   static vs direct DFT (7 bins) %.3e\n", maxref, e01, eref);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {1, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 200;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(NT), lds, 0, din, dout, dtwm, plan, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(NT), lds, 0, din, dout, dtws, plan, iters);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(NT), lds, 0, din, dout, dtws, plan, iters);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("grid %3d  %s  %.2f us per transform\n", grid, mode == 2 ? "static, tables in LDS" : mode ? "static 6x10x10x10   " : "mixed  4,4,5,5,5,3  ", ms * 1e3 / iters);
            }
        }
    }
    return 0;
}
