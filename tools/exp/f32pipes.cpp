// f32pipes.cpp — what the vector and matrix pipes of gfx950 sustain for float32, alone and side by side.  The numbers
// behind the decision to run two radix-8 passes of the N = 1024 STFT on the matrix cores (fft_core.h, MfmaFft512):
//   (1) operand / result layout of v_mfma_f32_16x16x4_f32 (printed as a mapping, checked against the hypothesis)
//   (2) v_fma_f32, v_add_f32, v_pk_fma_f32, v_log_f32 per-SIMD issue interval at 1..4 waves per SIMD
//   (3) v_mfma_f32_16x16x4_f32 issue interval, and VALU + MFMA streams from the same and from different waves
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/f32pipes tools/exp/f32pipes.cpp && tools/bin/f32pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}

// MODE: 0 v_fma_f32, 1 v_add_f32, 2 v_pk_fma_f32, 3 v_log_f32, 4 mfma only, 5 mfma + fma interleaved in one wave,
//       6 even waves mfma / odd waves fma, 7 v_fma_f64
template <int MODE>
__global__ void __launch_bounds__(512) pipe_kernel(float* out, int iters, long long* cycles) {
    float acc[16];
    f4 macc[4];
    const float x = (float)threadIdx.x * 1e-9f + 1.0f;
    double dacc[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)i;
#pragma unroll
    for (int i = 0; i < 4; ++i) macc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) dacc[i] = (double)i;
    const bool mfma_wave = ((threadIdx.x >> 8) & 1) == 0;     // 512-thread blocks: waves w and w + 4 share a SIMD
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(x));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(x));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 v = {acc[i], acc[i + 1]};
                f2 xx = {x, x};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(xx));
                acc[i] = v[0];
                acc[i + 1] = v[1];
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(acc[i]));
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, acc[i], macc[i], 0, 0, 0);
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    macc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, macc[i], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[(4 * i + q) & 15]) : "v"(x));
                }
        } else if (MODE == 6) {
            if (mfma_wave) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, macc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i]) : "v"(x));
            }
        } else if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dacc[i]) : "v"((double)x));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE>
static void run(const char* name, int insts_per_iter, float* out, long long* dcyc) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; ++wps) {                      // waves per SIMD: 256-thread blocks = 1 wave per SIMD each
        const int bs = MODE == 6 ? 512 : 256;
        if (MODE == 6 && (wps & 1)) continue;
        const int blocks = 256 * wps * 256 / bs;
        hipLaunchKernelGGL(pipe_kernel<MODE>, dim3(blocks), dim3(bs), 0, 0, out, 200, dcyc);      // warm
        HK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(pipe_kernel<MODE>, dim3(blocks), dim3(bs), 0, 0, out, iters, dcyc);
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc;
        HK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
        // per SIMD: wps waves, each iters * insts_per_iter instructions
        const double per_simd = (double)wps * iters * insts_per_iter;
        printf("%-34s %d waves/SIMD: %7.3f ms, wave-0 cycle counter %9lld -> %.2f counter-ticks per instruction per SIMD, %.2f ns per instruction per SIMD\n",
               name, wps, ms, cyc, (double)cyc / per_simd, ms * 1e6 / per_simd);
    }
}

int main() {
    float ha[64], hb[64], hd[256], *da, *db, *dd;
    HK(hipMalloc(&da, sizeof ha)); HK(hipMalloc(&db, sizeof hb)); HK(hipMalloc(&dd, sizeof hd));
    // hypothesis: lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = 4 (l / 16) + v][j = l % 16]
    for (int l = 0; l < 64; ++l) ha[l] = (float)((l % 16 + 1) * (l / 16 == 0 ? 1 : l / 16 == 1 ? 100 : l / 16 == 2 ? 10000 : 1000000));
    for (int l = 0; l < 64; ++l) hb[l] = (l % 16 == l / 16 + 3) ? 1.f : 0.f;     // B[k][j] = 1 iff j == k + 3 -> D[i][k + 3] = A[i][k]
    HK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); HK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dd);
    HK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
    int drow0 = 0, drow1 = 0, nz = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const float val = hd[l * 4 + v];
            if (val == 0.f) continue;
            ++nz;
            // decode A[i][k]: value = (i + 1) * 100^k
            int k = 0; float t = val;
            while (t > 99.5f) { t /= 100.f; ++k; }
            const int i = (int)(t + 0.5f) - 1;
            const int j = l % 16;
            if (j != k + 3) printf("unexpected column: lane %d v %d val %g\n", l, v, val);
            if (i == 4 * (l / 16) + v) ++drow0;
            if (i == 4 * v + l / 16) ++drow1;
            if (l < 64 && (l % 16 == 3)) printf("lane %2d v %d = %g  (row %d, k %d)\n", l, v, val, i, k);
        }
    printf("layout: %d non-zero results; D row = 4 (lane/16) + v matches %d, D row = 4 v + lane/16 matches %d  => FRT_MFMA_DROW=%d\n",
           nz, drow0, drow1, drow0 == nz ? 0 : drow1 == nz ? 1 : -1);

    float* out;
    long long* dcyc;
    HK(hipMalloc(&out, (size_t)256 * 4 * 256 * 4));
    HK(hipMalloc(&dcyc, 8));
    run<0>("v_fma_f32 x16", 16, out, dcyc);
    run<1>("v_add_f32 x16", 16, out, dcyc);
    run<2>("v_pk_fma_f32 x8", 8, out, dcyc);
    run<3>("v_log_f32 x16", 16, out, dcyc);
    run<7>("v_fma_f64 x16", 16, out, dcyc);
    run<4>("v_mfma_f32_16x16x4 x16", 16, out, dcyc);
    run<5>("16 mfma + 64 v_fma_f32, one wave", 80, out, dcyc);
    run<6>("even waves 16 mfma, odd 64 v_fma", 40, out, dcyc);
    return 0;
}
