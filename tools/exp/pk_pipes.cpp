// pk_pipes.cpp — round 4: throughput of the instruction classes stft_pk.h is made of, at the kernel's occupancy (one 512-thread
// workgroup per CU = two waves per SIMD) and at 1 / 4 waves per SIMD for reference.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/pk_pipes tools/exp/pk_pipes.cpp && tools/bin/pk_pipes
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
    f2 p[16];
    float a[16];
    const float x = (float)threadIdx.x * 1e-9f + 1.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { p[i] = f2{(float)i, x}; a[i] = (float)i + x; }
    const f2 xx = {x, x};
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long mask = 0x00ff00ff00ff00ffull + (unsigned long long)(iters & 1);
    const float vmask = __uint_as_float((lane & 8) ? 0xffffffffu : 0u);
    const unsigned ldsaddr = wave * 8192 + lane * 8;                      // wave-private 8 KB, conflict-free b64 accesses
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == 0) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(xx));
                REP16(X)
#undef X
            } else if (MODE == 1) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(xx));
                REP16(X)
#undef X
            } else if (MODE == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "+v"(p[i]) : "v"(xx));
                REP16(X)
#undef X
            } else if (MODE == 3) {          // two chains, alternating: a dependent instruction with one independent in between
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 1]) : "v"(xx));
                REP16(X)
#undef X
            } else if (MODE == 4) {          // four chains
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 3]) : "v"(xx));
                REP16(X)
#undef X
            } else if (MODE == 5) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i & 7]), "+v"(a[8 + (i & 7)]));
                REP16(X)
#undef X
            } else if (MODE == 6) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i & 7]), "+v"(a[8 + (i & 7)]));
                REP16(X)
#undef X
            } else if (MODE == 7) {
#define X(i) asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 5) & 15]) : "vcc");
                REP16(X)
#undef X
            } else if (MODE == 8) {          // 16 ds_write_b64 back to back (the transpose write / a scatter)
#define X(i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(ldsaddr), "v"(p[i]), "n"(i * 512) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 9) {          // 16 ds_read_b64
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[i]) : "v"(ldsaddr), "n"(i * 512) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 10) {         // one sub-transform pass of one round: 8 reads, 28 packed adds, 8 writes
#define X(i) if (i < 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[i]) : "v"(ldsaddr), "n"(i * 512) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 28; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(p[(j + 3) & 7]));
#define X(i) if (i < 8) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(ldsaddr), "v"(p[i]), "n"(i * 512) : "memory");
                REP16(X)
#undef X
            } else if (MODE == 12) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 5) & 15]));
                REP16(X)
#undef X
            } else if (MODE == 13) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 5) & 15]));
                REP16(X)
#undef X
            } else if (MODE == 14) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(a[(i + 5) & 15]) : "vcc");
                REP16(X)
#undef X
            } else if (MODE == 15) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(a[i]) : "v"(ldsaddr), "v"(a[(i + 5) & 15]) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 16) {
#define X(i) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM, \"00p00\")" : "=v"(a[i]) : "v"(a[(i + 5) & 15]) : "memory");
                REP16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 17) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
                REP16(X)
#undef X
            } else if (MODE == 18) {         // select on a mask held in a scalar register pair (VOP3 encoding)
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "s"(mask));
                REP16(X)
#undef X
            } else if (MODE == 19) {         // select on vcc, vcc only read (no clobber: no hazard nops from the compiler)
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
                REP16(X)
#undef X
            } else if (MODE == 20) {         // compare + select pairs
#define X(i) if (i & 1) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(a[(i + 5) & 15])); else asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(a[(i + 3) & 15]) : "vcc");
                REP16(X)
#undef X
            } else if (MODE == 21) {         // v_bfi_b32: a lane-mask select held in a vector register
#define X(i) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(vmask));
                REP16(X)
#undef X
            } else if (MODE == 11) {         // the same with the arithmetic only
#pragma unroll
                for (int j = 0; j < 28; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(p[(j + 3) & 7]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += p[i][0] + p[i][1] + a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 63];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* name, int instr_per_rep, float* out, unsigned long long* dst) {
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    const int wpss[] = {1, 2, 4};
    for (int wi = 0; wi < 3; ++wi) {
        const int wps = wpss[wi];
        const int bs = 256 * wps, blocks = 256;
        const size_t shm = 140 * 1024;                                   // one workgroup per CU
        HK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        const int iters = 20000 / wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), shm, 0, out, iters / 4, dst);
        HK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), shm, 0, out, iters, dst);
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2];
        HK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)st[0] / ((double)st[1] * 10.0);
        const double reps = (double)iters * 4;
        // whole-kernel time: what the SIMD (or the CU's LDS pipe) retires with every wave counted, not the favoured oldest one
        const double cyc_per_rep = ms * 1e6 * ghz / reps;
        printf("%-54s %d waves/SIMD: clock %.2f GHz  wave 0: %6.1f cycles per block of %2d = %5.2f / instr;  kernel: %6.2f cycles / instr / SIMD, %6.2f / instr / CU\n",
               name, wps, ghz, (double)st[0] / reps, instr_per_rep, (double)st[0] / reps / instr_per_rep, cyc_per_rep / (instr_per_rep * wps), cyc_per_rep / (instr_per_rep * wps * 4));
    }
}

int main() {
    float* out;
    unsigned long long* dst;
    HK(hipMalloc(&out, (size_t)512 * 1024 * 4));
    HK(hipMalloc(&dst, 16));
    {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        HK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        for (;;) {
            hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 140 * 1024, 0, out, 2000, dst);
            HK(hipDeviceSynchronize());
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 >= 300.0) break;
        }
    }
    run<0>("v_pk_add_f32, 16 chains", 16, out, dst);
    if (getenv("PK_PIPES_SELECT")) goto selects;
    run<1>("v_pk_mul_f32, 16 chains", 16, out, dst);
    run<2>("v_pk_add_f32 with op_sel / neg, 16 chains", 16, out, dst);
    run<4>("v_pk_add_f32, 4 chains", 16, out, dst);
    run<3>("v_pk_add_f32, 2 chains", 16, out, dst);
    run<5>("v_permlane32_swap_b32", 16, out, dst);
    run<6>("v_permlane16_swap_b32", 16, out, dst);
    run<7>("v_cndmask_b32_dpp row_ror:8", 16, out, dst);
    run<12>("v_mov_b32_dpp row_ror:8", 16, out, dst);
    run<13>("v_mov_b32_dpp quad_perm", 16, out, dst);
    run<17>("v_add_f32_dpp row_ror:8", 16, out, dst);
selects:
    run<14>("v_cndmask_b32 (vcc)", 16, out, dst);
    run<18>("v_cndmask_b32_e64 (sgpr pair mask)", 16, out, dst);
    run<19>("v_cndmask_b32 (vcc read only)", 16, out, dst);
    run<20>("8 v_cmp_gt_f32 + 8 v_cndmask_b32", 16, out, dst);
    run<21>("v_bfi_b32 (vector mask select)", 16, out, dst);
    return 0;
    run<15>("16 ds_bpermute_b32 + wait", 16, out, dst);
    run<16>("16 ds_swizzle_b32 + wait", 16, out, dst);
    run<8>("16 ds_write_b64 + wait", 16, out, dst);
    run<9>("16 ds_read_b64 + wait", 16, out, dst);
    run<10>("8 ds_read_b64, wait, 28 v_pk_add_f32, 8 ds_write_b64", 44, out, dst);
    run<11>("28 v_pk_add_f32 (dependent at distance 3)", 28, out, dst);
    return 0;
}
