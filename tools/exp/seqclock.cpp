// seqclock.cpp — what bounds a single-wavefront sequential recurrence (the bit-exact IIR stage of the streaming chains)?
// Times dependent chains on ONE wave: f64 fma chain, fma + DPP row broadcast (the IIR step's critical path), with and
// without a saturating background kernel on another stream (does the shader clock ramp for a one-wave kernel?).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/seqclock tools/exp/seqclock.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ double bcast0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void __launch_bounds__(64) chain(double* out, int n, double a, double b, long long* clk) {
    double z = threadIdx.x * 1e-3, acc = 0;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) z = __builtin_fma(z, a, b);                       // plain dependent fma chain
            if (MODE == 1) {                                                   // the fused IIR step
                const double x = b + u;
                const double y = bcast0(__builtin_fma(a, x, z));
                const double zn = shl1(z);
                z = __builtin_fma(-y, a, __builtin_fma(x, b, zn));
                acc += y;
            }
            if (MODE == 2) {                                                   // the exact (separately rounded) step
                const double x = b + u;
                const double y = bcast0(z + a * x);
                const double zn = shl1(z);
                z = __builtin_fma(zn, 1.0, x * b) - y * a;
                acc += y;
            }
            if (MODE == 3) {                                                   // readlane instead of DPP broadcast
                const double x = b + u;
                const double yv = z + a * x;
                const double y = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(yv)), __builtin_amdgcn_readfirstlane(__double2loint(yv)));
                const double zn = shl1(z);
                z = __builtin_fma(zn, 1.0, x * b) - y * a;
                acc += y;
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = z + acc;
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

__global__ void busy(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

template <int MODE>
int run(const char* name, int n, bool background) {
    double* out; long long* clk; float* bp;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&clk, 16)); CK(hipMalloc(&bp, 4096 * 256 * 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        usleep(rep == 0 ? 200000 : 0);
        if (background) hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, s2, bp, 2000000);
        CK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, s1, out, n, 0.999, 1e-3, clk);
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        printf("%-28s bg=%d rep=%d  n=%d  %.1f us  %.1f ns/step  clock64 %.1f/step  wall_clock64 %.2f/step\n", name, background, rep, n, ms * 1e3,
               ms * 1e6 / n, (double)h[0] / n, (double)h[1] / n);
        CK(hipDeviceSynchronize());
    }
    return 0;
}

int main() {
    for (int bg = 0; bg < 2; ++bg) {
        run<0>("fma chain", 65536, bg);
        run<1>("fused iir step", 65536, bg);
        run<2>("exact iir step", 65536, bg);
        run<3>("exact step, readfirstlane", 65536, bg);
        run<2>("exact iir step, 512", 512, bg);
    }
    return 0;
}
