// storeshape.cpp — what the row stores of the N = 1024 STFT kernel cost by shape.  Every wavefront writes `run` consecutive
// rows of 513 dwords (2052 bytes), no loads, no arithmetic:
//   0  the kernel's shape: four lane-ascending 256-byte dword stores, four lane-DESCENDING ones, one single-lane store
//   1  eight ascending dword stores + one single-lane store
//   2  the same bytes as dwordx2 stores (lane holds two adjacent bins) + tail
//   3  the same bytes as dwordx4 stores, row treated as part of the run's contiguous byte range, 16-byte aligned chunks
//   4  as 3, nontemporal
//   5  as 1, nontemporal
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/storeshape tools/exp/storeshape.cpp && tools/bin/storeshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <time.h>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256, 3) k(float* out, long long n_rows, int run, int spin) {
    const int lane = threadIdx.x & 63;
    const long long grp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long f0 = grp * run;
    const float v = (float)lane;
    if (MODE <= 2 || MODE == 5 || MODE >= 6) {   // (13 included)
        for (int g = 0; g < run; ++g) {
            const long long f = f0 + g;
            if (f >= n_rows) break;
            float* row = out + f * 513;
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    row[lane + 64 * j] = v;
                    row[512 - lane - 64 * j] = v;
                }
                if (lane == 0) row[256] = v;
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) row[lane + 64 * j] = v;
                if (lane == 0) row[512] = v;
            } else if (MODE == 13) {
                // the kernel's nine stores of a row SPREAD through the arithmetic (one store, a ninth of the multiply-adds, ...)
                // instead of issued back to back
                float acc = v;
                const int part = spin / 9;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    if (j < 4) row[lane + 64 * j] = v;
                    else if (j < 8) row[512 - lane - 64 * (j - 4)] = v;
                    else if (lane == 0) row[256] = v;
                    for (int t = 0; t < part; ++t) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
                }
                if (acc == 12345.f) row[0] = acc;
            } else if (MODE >= 10) {
                // two 16-byte-per-lane stores per row (2048 of its 2052 bytes, alignment ignored: timing only) + the arithmetic
                f4* p4 = (f4*)(out + (f * 513 & ~3ll)) + lane;
                if (MODE == 11) { __builtin_nontemporal_store(f4{v, v, v, v}, p4); __builtin_nontemporal_store(f4{v, v, v, v}, p4 + 64); }
                else { p4[0] = f4{v, v, v, v}; p4[64] = f4{v, v, v, v}; }
                float acc = v;
                for (int t = 0; t < spin; ++t) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
                if (acc == 12345.f) row[0] = acc;
            } else if (MODE >= 9) {
                // the kernel's shape with a scalar row base + 32-bit lane offsets (global_store_dword v_off, v_data, s[base])
                float* srow = (float*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)row >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)row));
                const unsigned lo = lane, hi = 512 - lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    srow[lo + 64 * j] = v;
                    srow[hi - 64 * j] = v;
                }
                if (lane == 0) srow[256] = v;
                float acc = v;
                for (int t = 0; t < spin; ++t) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
                if (acc == 12345.f) row[0] = acc;
            } else if (MODE >= 6) {
                // the kernel's shape, throttled the way the frame loop throttles it: the wait for the next frame's samples
                // (vmcnt counts in order) also waits for the row stores issued before them.  MODE 6: every row acknowledged
                // before the next is stored; 7: one row may still be in flight; 8: two rows
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    row[lane + 64 * j] = v;
                    row[512 - lane - 64 * j] = v;
                }
                if (lane == 0) row[256] = v;
                if (MODE == 6) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
                else if (MODE == 7) __builtin_amdgcn_s_waitcnt(0x0F79);     // vmcnt(9)
                else __builtin_amdgcn_s_waitcnt(0x4F72);                    // vmcnt(18)
                // ~ a frame's worth of arithmetic between rows
                float acc = v;
                for (int t = 0; t < spin; ++t) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
                if (acc == 12345.f) row[0] = acc;
            } else if (MODE == 5) {
#pragma unroll
                for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(v, row + lane + 64 * j);
                if (lane == 0) __builtin_nontemporal_store(v, row + 512);
            } else {
                // dwordx2 needs 8-byte alignment: rows start on 4-byte boundaries -> odd rows shift by one
                const int sh = (int)(f & 1);
                if (lane == 0 && sh) row[0] = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f2*)(row + sh + 2 * lane + 128 * j) = f2{v, v};
                if (lane == 0 && !sh) row[512] = v;
            }
        }
    } else {
        // the run's rows are one contiguous byte range: aligned 1 KB chunks (64 lanes x 16 bytes), dword head and tail
        long long nr = n_rows - f0;
        if (nr > run) nr = run;
        if (nr <= 0) return;
        float* base = out + f0 * 513;
        const long long total = nr * 513;                     // dwords
        const long long head = ((16 - ((uintptr_t)base & 15)) & 15) / 4;
        if (lane < head) base[lane] = v;
        const long long nchunk = (total - head) / 256;
        for (long long c = 0; c < nchunk; ++c) {
            f4* p = (f4*)(base + head + c * 256) + lane;
            if (MODE == 4) __builtin_nontemporal_store(f4{v, v, v, v}, p);
            else *p = f4{v, v, v, v};
        }
        for (long long t = head + nchunk * 256 + lane; t < total; t += 64) base[t] = v;
    }
}

// producer / consumer split: 320-thread workgroups, wavefronts 0-3 do the arithmetic of their rows and leave the row in an LDS
// slot (nine ds_write_b32), wavefront 4 takes the four rows of the previous step from LDS and stores them (dwordx4 where the
// alignment allows): one workgroup barrier per step, double-buffered slots.  Same occupancy as the kernel (three workgroups
// per CU: launch bound and LDS as there).
__global__ void __launch_bounds__(320, 3) k_split(float* out, long long n_rows, int run, int spin) {
    __shared__ float slot[2][4][516];
    __shared__ float pad[4608];                                   // the transform's exchange buffers of the real kernel (occupancy)
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const float v = (float)lane;
    if (spin < 0) pad[threadIdx.x] = v;
    float acc = v;
    for (int g = 0; g <= run; ++g) {
        if (w < 4) {
            if (g < run) {
                for (int t = 0; t < spin; ++t) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
                float* r = slot[g & 1][w];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r[lane + 64 * j] = acc;
                    r[512 - lane - 64 * j] = acc;
                }
                if (lane == 0) r[256] = acc;
            }
        } else if (g > 0) {
            for (int ww = 0; ww < 4; ++ww) {
                const long long f = ((long long)blockIdx.x * 4 + ww) * run + (g - 1);
                if (f >= n_rows) continue;
                float* row = out + f * 513;
                const float* r = slot[(g - 1) & 1][ww];
#pragma unroll
                for (int j = 0; j < 8; ++j) row[lane + 64 * j] = r[lane + 64 * j];
                if (lane == 0) row[512] = r[512];
            }
        }
        __syncthreads();
    }
    if (acc == 12345.f) out[0] = acc;
}

static void run_split(float* out, long long n_rows, int run, int spin) {
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    const long long groups = (n_rows + run - 1) / run;
    const int blocks = (int)((groups + 3) / 4);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_split, dim3(blocks), dim3(320), 0, 0, out, n_rows, run, spin);
    HK(hipEventRecord(e0, 0));
    const int iters = 40;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_split, dim3(blocks), dim3(320), 0, 0, out + (size_t)(i & 3) * n_rows * 513, n_rows, run, spin);
    HK(hipEventRecord(e1, 0));
    HK(hipEventSynchronize(e1));
    float ms;
    HK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("%-60s run=%2d spin=%3d  %.4f ms  %.0f GB/s written\n", "12 four arithmetic wavefronts + one storing wavefront (LDS hand-off)", run, spin, ms, n_rows * 2052.0 / ms * 1e-6);
}

template <int MODE>
static void run_mode(const char* name, float* out, long long n_rows, int run, int spin = 0) {
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    const long long groups = (n_rows + run - 1) / run;
    const int blocks = (int)((groups + 3) / 4);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, n_rows, run, spin);
    HK(hipEventRecord(e0, 0));
    const int iters = 40;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out + (size_t)(i & 3) * n_rows * 513, n_rows, run, spin);
    HK(hipEventRecord(e1, 0));
    HK(hipEventSynchronize(e1));
    float ms;
    HK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("%-60s run=%2d spin=%3d  %.4f ms  %.0f GB/s written\n", name, run, spin, ms, n_rows * 2052.0 / ms * 1e-6);
}

int main() {
    const long long n_rows = 131071;
    float* out;
    HK(hipMalloc(&out, (size_t)4 * n_rows * 513 * 4 + 4096));
    {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, n_rows, 16, 0);
            HK(hipDeviceSynchronize());
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 >= 300.0) break;
        }
    }
    for (int run : {16, 8}) {
        run_mode<0>("0 kernel shape: 4 ascending + 4 descending dword + 1", out, n_rows, run);
        run_mode<1>("1 eight ascending dword stores + 1", out, n_rows, run);
        run_mode<5>("5 eight ascending dword stores + 1, nontemporal", out, n_rows, run);
        run_mode<2>("2 dwordx2 stores", out, n_rows, run);
        run_mode<3>("3 aligned dwordx4 chunks of the run's byte range", out, n_rows, run);
        run_mode<4>("4 aligned dwordx4 chunks, nontemporal", out, n_rows, run);
    }
    for (int spin : {0, 100, 200, 300}) run_split(out, n_rows, 16, spin);
    for (int spin : {90, 180, 270}) {
        run_mode<13>("13 kernel shape, stores spread through the arithmetic", out, n_rows, 16, spin);
        run_mode<7>("7 kernel shape, stores back to back", out, n_rows, 16, spin);
    }
    for (int spin : {0, 100, 200}) {
        run_mode<10>("10 two dwordx4 stores per row", out, n_rows, 16, spin);
        run_mode<11>("11 two dwordx4 stores per row, nontemporal", out, n_rows, 16, spin);
        run_mode<9>("9 kernel shape, scalar base + lane offset", out, n_rows, 16, spin);
        run_mode<7>("7 kernel shape, vmcnt(9): one row in flight", out, n_rows, 16, spin);
    }
    for (int spin : {300}) {
        run_mode<6>("6 kernel shape, vmcnt(0) after every row", out, n_rows, 16, spin);
        run_mode<7>("7 kernel shape, vmcnt(9): one row in flight", out, n_rows, 16, spin);
        run_mode<8>("8 kernel shape, vmcnt(18): two rows in flight", out, n_rows, 16, spin);
    }
    return 0;
}
