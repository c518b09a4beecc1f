// f64_pipes.cpp — round 4: issue rates of the float64 instructions the overlap-add bank is made of, per SIMD at 1 / 2 / 4 waves.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/f64_pipes tools/exp/f64_pipes.cpp && tools/bin/f64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(double* out, int iters, unsigned long long* stamps) {
    extern __shared__ double lds[];
    double a[16];
    d2 q[16];
    const double x = (double)threadIdx.x * 1e-12 + 1.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = (double)i + x; q[i] = d2{a[i], x}; }
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned ldsaddr = wave * 8192 + lane * 16;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == 0) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 3) {          // 4 chains
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i & 3]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 4) {          // 2 chains
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i & 1]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 5) {          // mixed: add, mul, fma alternating
#define X(i) if (i % 3 == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x)); else if (i % 3 == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x)); else asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x));
                REP16(X)
#undef X
            } else if (MODE == 6) {
#define X(i) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(ldsaddr), "v"(q[i]), "n"(i * 1024 % 8192) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 7) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(ldsaddr), "n"(i * 1024 % 8192) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 8) {          // add with a scalar-register constant operand
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "s"(1.0000001));
                REP16(X)
#undef X
            } else if (MODE == 9) {          // f32 reference
                float* f = (float*)a;
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"((float)x));
                REP16(X)
#undef X
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + q[i][0] + q[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 63];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* name, double* out, unsigned long long* dst) {
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    const int wpss[] = {1, 2, 4};
    for (int wi = 0; wi < 3; ++wi) {
        const int wps = wpss[wi], bs = 256 * wps, blocks = 256;
        const size_t shm = 140 * 1024;                                   // one workgroup per CU
        HK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        const int iters = 8000 / wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), shm, 0, out, iters / 4, dst);
        HK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), shm, 0, out, iters, dst);
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2];
        HK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)st[0] / ((double)st[1] * 10.0), reps = (double)iters * 4;
        const double cyc_per_rep = ms * 1e6 * ghz / reps;
        printf("%-40s %d waves/SIMD: clock %.2f GHz  wave 0: %5.2f cycles / instr;  whole kernel: %6.2f cycles / instr / SIMD\n", name, wps, ghz,
               (double)st[0] / reps / 16, cyc_per_rep / (16 * wps));
    }
}

int main() {
    double* out;
    unsigned long long* dst;
    HK(hipMalloc(&out, (size_t)512 * 1024 * 8));
    HK(hipMalloc(&dst, 16));
    run<9>("v_fma_f32, 16 chains (warm-up)", out, dst);
    run<9>("v_fma_f32, 16 chains", out, dst);
    run<0>("v_add_f64, 16 chains", out, dst);
    run<1>("v_mul_f64, 16 chains", out, dst);
    run<2>("v_fma_f64, 16 chains", out, dst);
    run<5>("add / mul / fma f64 alternating", out, dst);
    run<3>("v_add_f64, 4 chains", out, dst);
    run<4>("v_add_f64, 2 chains", out, dst);
    run<8>("v_add_f64 with a scalar operand", out, dst);
    run<6>("16 ds_write_b128 + wait", out, dst);
    run<7>("16 ds_read_b128 + wait", out, dst);
    return 0;
}
