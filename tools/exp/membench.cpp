// membench.cpp — what the memory pipeline sustains for the STFT kernel's access shape (no compute).
// Every wave walks `run` consecutive frames: reads 2 KB of new samples per frame, writes a 2052-byte row.
//   mode 0: 4 x dwordx2 loads, 9 x dword stores per lane-frame (what stft_kernel issues)
//   mode 1: 2 x dwordx4 loads, 9 x dword stores
//   mode 2: 4 x dwordx2 loads, 2 x dwordx4 (+1 dword) stores to the 4-byte aligned row
//   mode 3: 2 x dwordx4 loads, 2 x dwordx4 (+1 dword) stores
//   mode 4: as 0 with the loads issued two frames ahead
//   mode 5: as 0 with rows padded to 528 words (64-byte aligned rows)
//   mode 6: as 0 with non-temporal stores;  mode 7: non-temporal loads and stores
//   mode 8: plain streaming copy of the same byte counts (dwordx4, fully linear)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>
__global__ void __launch_bounds__(256) mem_kernel(const float* __restrict__ x, float* __restrict__ out, int run, int nframes) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const long long f0 = (long long)wave * run;
    if (f0 >= nframes) return;
    const float* xs = x + f0 * 512 + 512;          // second half of frame f0 onwards
    constexpr int RS = MODE == 5 ? 528 : 513;
    float* row = out + f0 * RS;
    float acc = 0.f;
    float2 a[4], n1[4], n2[4];
    float4 b[2];
    if (MODE == 4) {
        for (int j = 0; j < 4; ++j) n1[j] = ((const float2*)xs)[lane + 64 * j];
        for (int j = 0; j < 4; ++j) n2[j] = ((const float2*)(xs + 512))[lane + 64 * j];
    }
    for (int g = 0; g < run && f0 + g < nframes; ++g) {
        float v[8];
        if (MODE == 7) {
            for (int j = 0; j < 4; ++j) {
                a[j].x = __builtin_nontemporal_load(xs + (long long)g * 512 + 2 * (lane + 64 * j));
                a[j].y = __builtin_nontemporal_load(xs + (long long)g * 512 + 2 * (lane + 64 * j) + 1);
            }
            for (int j = 0; j < 4; ++j) { v[2 * j] = a[j].x + acc; v[2 * j + 1] = a[j].y; }
        } else if (MODE == 0 || MODE == 2 || MODE == 5 || MODE == 6) {
            for (int j = 0; j < 4; ++j) a[j] = ((const float2*)(xs + (long long)g * 512))[lane + 64 * j];
            for (int j = 0; j < 4; ++j) { v[2 * j] = a[j].x + acc; v[2 * j + 1] = a[j].y; }
        } else if (MODE == 4) {
            for (int j = 0; j < 4; ++j) a[j] = n1[j];
            for (int j = 0; j < 4; ++j) n1[j] = n2[j];
            if (g + 2 < run) for (int j = 0; j < 4; ++j) n2[j] = ((const float2*)(xs + (long long)(g + 2) * 512))[lane + 64 * j];
            for (int j = 0; j < 4; ++j) { v[2 * j] = a[j].x + acc; v[2 * j + 1] = a[j].y; }
        } else {
            for (int j = 0; j < 2; ++j) b[j] = ((const float4*)(xs + (long long)g * 512))[lane + 64 * j];
            for (int j = 0; j < 2; ++j) { v[4 * j] = b[j].x + acc; v[4 * j + 1] = b[j].y; v[4 * j + 2] = b[j].z; v[4 * j + 3] = b[j].w; }
        }
        acc = v[7] * 1e-9f;
        float* r = row + (long long)g * RS;
        if (MODE == 6 || MODE == 7) {
            for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(v[j], r + lane + 64 * j);
            if (lane == 0) r[512] = v[0];
        } else if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5) {
            for (int j = 0; j < 8; ++j) r[lane + 64 * j] = v[j];
            if (lane == 0) r[512] = v[0];
        } else {
            *(f4u*)(r + 4 * lane) = f4u{v[0], v[1], v[2], v[3]};
            *(f4u*)(r + 256 + 4 * lane) = f4u{v[4], v[5], v[6], v[7]};
            if (lane == 0) r[512] = v[0];
        }
    }
}

__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ x, float4* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = x[i];
}

int main(int argc, char** argv) {
    // argv[2]: number of distinct input/output buffer sets the launches rotate over (1 = the same 268 MB input
    // every launch, which the 256 MB Infinity Cache can partly retain; 4 = every launch sees cold data)
    const int nframes = 131071, run = argc > 1 ? atoi(argv[1]) : 16, sets = argc > 2 ? atoi(argv[2]) : 1;
    float *xb, *outb;
    const size_t xn = (size_t)(nframes + 2) * 512, on = (size_t)nframes * 528;
    HK(hipMalloc(&xb, xn * 4 * sets));
    HK(hipMalloc(&outb, on * 4 * sets));
    HK(hipMemset(xb, 0, xn * 4 * sets));
    int it = 0;
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    const int waves = (nframes + run - 1) / run, blocks = (waves + 3) / 4;
    for (int mode = 0; mode < 9; ++mode) {
        auto launch = [&]() {
            float* x = xb + xn * (it % sets);
            float* out = outb + on * (it % sets);
            ++it;
            switch (mode) {
                case 0: hipLaunchKernelGGL(mem_kernel<0>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 1: hipLaunchKernelGGL(mem_kernel<1>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 2: hipLaunchKernelGGL(mem_kernel<2>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 3: hipLaunchKernelGGL(mem_kernel<3>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 4: hipLaunchKernelGGL(mem_kernel<4>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 5: hipLaunchKernelGGL(mem_kernel<5>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 6: hipLaunchKernelGGL(mem_kernel<6>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                case 7: hipLaunchKernelGGL(mem_kernel<7>, dim3(blocks), dim3(256), 0, 0, x, out, run, nframes); break;
                default: hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, (const float4*)x, (float4*)out, (long long)nframes * 128); break;
            }
        };
        for (int i = 0; i < (mode == 0 ? 3000 : 300); ++i) launch();      // clock ramp: reach the sustained clocks before timing
        HK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; ++i) launch();
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / 200 * 1e-3, bytes = (double)nframes * 4100;
        printf("mode %d run %d sets %d: %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)\n", mode, run, sets, per * 1e3, bytes / per * 1e-9, bytes / per / 8e10);
    }
    return 0;
}
