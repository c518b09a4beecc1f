// issue_model.cpp — per-wave issue interval and per-SIMD throughput of the instruction classes the STFT kernels are made
// of, at 1..8 waves per SIMD and sustained clocks.  One workgroup per CU (LDS reservation), waves per SIMD = block / 256.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/issue_model tools/exp/issue_model.cpp && tools/bin/issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <time.h>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters, unsigned long long* stamps) {
    extern __shared__ float lds[];
    float a[16];
    f2 p[8];
    const float x = (float)threadIdx.x * 1e-9f + 1.0f;
    int s0 = iters, s1 = 1, s2 = 2, s3 = 3;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)i + x;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{(float)i, x};
    const f2 xx = {x, x};
    const unsigned ldsaddr = (threadIdx.x & 63) * 4;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 1) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(xx));
                REP16(X)
#undef X
            } else if (MODE == 4) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                REP16(X)
#undef X
            } else if (MODE == 5) {
#define X(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc");
                REP16(X)
#undef X
            } else if (MODE == 6) {          // 8 VALU + 8 SALU interleaved, independent
#define X(i) if (i & 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s2) : "scc"); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 7) {          // 12 VALU + 4 ds_read_b32 (results unused until the end of the block)
#define X(i) if ((i & 3) == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(ldsaddr)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (MODE == 8) {          // dependent pairs: 8 chains of 2
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i >> 1]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 9) {          // 4 chains
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i & 3]) : "v"(x));
                REP16(X)
#undef X
            }
        }
    }
    float s = (float)(s1 + s3 + s0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 63];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* dst) {
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    const int wpss[] = {1, 2, 3, 4, 8};
    for (int wi = 0; wi < 5; ++wi) {
        const int wps = wpss[wi];
        const int bs = wps == 8 ? 1024 : 256 * wps;
        const int blocks = wps == 8 ? 512 : 256;
        const size_t shm = wps == 8 ? 70 * 1024 : 100 * 1024;          // 1 (or 2) workgroups per CU
        HK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        const int iters = 60000 / wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), shm, 0, out, iters / 4, dst);
        HK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), shm, 0, out, iters, dst);
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2];
        HK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)st[0] / ((double)st[1] * 10.0);      // s_memrealtime: 100 MHz
        const double per_simd = (double)wps * iters * 64;
        printf("%-44s %d waves/SIMD: %7.3f ms  clock %.2f GHz (memtime/memrealtime)  %.2f ns = %.2f cycles per instruction per SIMD; wave 0: %.2f memtime ticks per instruction\n",
               name, wps, ms, ghz, ms * 1e6 / per_simd, ms * 1e6 / per_simd * ghz, (double)st[0] / ((double)iters * 64));
    }
}

int main() {
    float* out;
    unsigned long long* dst;
    HK(hipMalloc(&out, (size_t)512 * 1024 * 4));
    HK(hipMalloc(&dst, 16));
    {   // sustained clocks first
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        HK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        for (;;) {
            hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 100 * 1024, 0, out, 2000, dst);
            HK(hipDeviceSynchronize());
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 >= 400.0) break;
        }
    }
    run<0>("v_add_f32, 16 independent chains", out, dst);
    run<9>("v_add_f32, 4 chains", out, dst);
    run<8>("v_add_f32, 8 chains of dependent pairs", out, dst);
    run<1>("v_add_f32, one dependent chain", out, dst);
    run<2>("v_fma_f32, 16 chains", out, dst);
    run<3>("v_pk_fma_f32, 8 chains", out, dst);
    run<4>("v_log_f32, 16 chains", out, dst);
    run<5>("s_add_u32, one chain", out, dst);
    run<6>("8 v_add_f32 + 8 s_add_u32 interleaved", out, dst);
    run<7>("12 v_add_f32 + 4 ds_read_b32", out, dst);
    return 0;
}
