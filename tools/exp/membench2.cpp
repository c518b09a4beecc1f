// membench2.cpp — where is this chip's streaming ceiling?  (round 2: reconcile the 5.1 TB/s of membench.cpp with the
// 6.29 TB/s float4 copy of MI355X_MICROARCH.md.)
//
//   membench2 [MiB per buffer = 256] [buffer sets = 4]
//
// Sections, each line "tag ... GB/s" counts bytes read + bytes written:
//   memcpy   hipMemcpyAsync device-to-device (the runtime's own blit kernel)
//   copy     float4 copy, U loads in flight per thread before the first store, grids of 256 x {4, 8, 16} blocks,
//            cache policy plain / nt loads + nt stores / nt stores only
//   read     read-only stream (sum), U loads in flight
//   write    write-only stream (fill)
//   stft     the STFT kernel's shape (2 KB read + 2052-byte row written per wave and frame) at 12 or 32 waves per CU,
//            prefetch depth 0 / 1 / 2 / 3 frames, 513- or 516-word rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define HK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int POLICY> __device__ __forceinline__ f4 ld(const f4* p) {
    if (POLICY == 1 || POLICY == 3) return __builtin_nontemporal_load(p);
    return *p;
}
template <int POLICY> __device__ __forceinline__ void st(f4* p, f4 v) {
    if (POLICY == 1 || POLICY == 2) __builtin_nontemporal_store(v, p); else *p = v;
}

// block-contiguous chunks of 256*U float4; grid-stride over chunks
template <int U, int POLICY>
__global__ void __launch_bounds__(256) copy_k(const f4* __restrict__ x, f4* __restrict__ out, long long nchunks) {
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const f4* s = x + c * (256 * U) + threadIdx.x;
        f4* d = out + c * (256 * U) + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = ld<POLICY>(s + 256 * j);
#pragma unroll
        for (int j = 0; j < U; ++j) st<POLICY>(d + 256 * j, v[j]);
    }
}

template <int U>
__global__ void __launch_bounds__(256) read_k(const f4* __restrict__ x, float* __restrict__ out, long long nchunks) {
    f4 acc = {0, 0, 0, 0};
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const f4* s = x + c * (256 * U) + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = s[256 * j];
#pragma unroll
        for (int j = 0; j < U; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x;
}

template <int U>
__global__ void __launch_bounds__(256) write_k(f4* __restrict__ out, long long nchunks, float val) {
    const f4 v = {val, val, val, val};
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        f4* d = out + c * (256 * U) + threadIdx.x;
#pragma unroll
        for (int j = 0; j < U; ++j) d[256 * j] = v;
    }
}

// STFT shape.  DEPTH frames of loads in flight ahead of the frame being stored; LDSB bytes of LDS per block cap the
// resident blocks per CU (53248 -> 3 blocks = 12 waves per CU like stft_kernel's 3 waves per SIMD); RS words per row;
// POLICY as above (1 nt loads + nt stores, 2 nt stores, 3 nt loads);
// STORE 0: nine dword stores per lane and row (what stft_kernel issues)
// STORE 1: the wave's rows form one contiguous word stream: rows are appended to a 1024-word LDS ring and leave it as
//          1 KB chunks on 1 KB boundaries, one aligned dwordx4 per lane (head and tail of the run: dword stores)
template <int DEPTH, int LDSB, int RS, int POLICY, int STORE>
__global__ void __launch_bounds__(256) stft_shape_k(const float* __restrict__ x, float* __restrict__ out, int run, int nframes) {
    __shared__ float pad[LDSB / 4 > 0 ? LDSB / 4 : 1];
    __shared__ __attribute__((aligned(16))) float ring[STORE ? 4 * 1024 : 1];
    if (LDSB > 0 && run < 0) pad[threadIdx.x] = 1.f;            // keeps the allocation
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const long long f0 = (long long)wave * run;
    if (f0 >= nframes) return;
    const float* xs = x + f0 * 512 + 512;
    float* row = out + f0 * RS;
    constexpr int D = DEPTH < 1 ? 1 : DEPTH;
    float2 q[D][4];
    long long nfr = nframes - f0;
    if (nfr > run) nfr = run;
    auto load2 = [&](const float* p) -> float2 {
        if (POLICY == 1 || POLICY == 3) {
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 t = __builtin_nontemporal_load((const v2*)p);
            return {t.x, t.y};
        }
        return *(const float2*)p;
    };
    if (DEPTH >= 1) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nfr)
#pragma unroll
                for (int j = 0; j < 4; ++j) q[d][j] = load2(xs + (long long)d * 512 + 2 * (lane + 64 * j));
    }
    float acc = 0.f;
    float* wring = ring + (threadIdx.x >> 6) * 1024;
    const long long s0 = f0 * RS;                          // first word of the wave's stream
    long long done = (s0 + 255) & ~255ll;                  // words below this have left (or belong to the head)
    for (int g = 0; g < nfr; ++g) {
        float2 a[4];
        if (DEPTH == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = load2(xs + (long long)g * 512 + 2 * (lane + 64 * j));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = q[0][j];
#pragma unroll
            for (int d = 0; d + 1 < D; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) q[d][j] = q[d + 1][j];
            if (g + D < nfr)
#pragma unroll
                for (int j = 0; j < 4; ++j) q[D - 1][j] = load2(xs + (long long)(g + D) * 512 + 2 * (lane + 64 * j));
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = a[j].x + acc; v[2 * j + 1] = a[j].y; }
        acc = v[7] * 1e-9f;
        if (STORE == 0) {
            float* r = row + (long long)g * RS;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (POLICY == 1 || POLICY == 2) __builtin_nontemporal_store(v[j], r + lane + 64 * j);
                else r[lane + 64 * j] = v[j];
            }
            if (lane == 0) r[512] = v[0];
        } else {
            const long long w0 = s0 + (long long)g * RS;  // stream position of this row
            if (w0 < done) {                               // head of the run: words below the first 1 KB boundary
#pragma unroll
                for (int j = 0; j < 8; ++j) if (w0 + lane + 64 * j < done) out[w0 + lane + 64 * j] = v[j];
                if (lane == 0 && w0 + 512 < done) out[w0 + 512] = v[0];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) wring[(w0 + lane + 64 * j) & 1023] = v[j];
            if (lane == 0) wring[(w0 + 512) & 1023] = v[0];
            const long long avail = w0 + RS;
            while (done + 256 <= avail) {                  // wave-uniform: 2 chunks per row, sometimes 3
                const f4 c = *(const f4*)(wring + ((done + 4 * lane) & 1023));
                if (POLICY == 1 || POLICY == 2) __builtin_nontemporal_store(c, (f4*)(out + done) + lane);
                else *((f4*)(out + done) + lane) = c;
                done += 256;
            }
            if (g + 1 == nfr) {                            // tail of the run
                for (long long w = done + lane; w < avail; w += 64) out[w] = wring[w & 1023];
            }
        }
    }
    if (LDSB > 0 && acc == 12345.f) out[0] = pad[lane];
}

static hipEvent_t e0, e1;
static double time_ms(const std::function<void(int)>& launch, int warm, int iters) {
    for (int i = 0; i < warm; ++i) launch(i);
    HK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch(i);
    HK(hipEventRecord(e1, 0));
    HK(hipEventSynchronize(e1));
    float ms;
    HK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 256;
    const int sets = argc > 2 ? atoi(argv[2]) : 4;
    const size_t bytes = mib << 20, n4 = bytes / 16;
    char *xb, *ob;
    HK(hipMalloc(&xb, bytes * sets + (1 << 20)));
    HK(hipMalloc(&ob, (bytes + (bytes >> 4)) * sets + (1 << 20)));
    HK(hipMemset(xb, 0, bytes * sets));
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    auto X = [&](int i) { return (f4*)(xb + bytes * (i % sets)); };
    auto O = [&](int i) { return (f4*)(ob + (bytes + (bytes >> 4)) * (i % sets)); };
    auto report = [&](const char* tag, double ms, double moved) {
        printf("%-44s %8.4f ms %7.0f GB/s (%.1f%% of 8 TB/s)\n", tag, ms, moved / ms * 1e-6, moved / ms * 1e-6 / 80.0);
        fflush(stdout);
    };
    // clock ramp
    time_ms([&](int i) { hipLaunchKernelGGL((copy_k<4, 0>), dim3(2048), dim3(256), 0, 0, X(i), O(i), (long long)(n4 / 1024)); }, 200, 2000);

    report("memcpy d2d", time_ms([&](int i) { HK(hipMemcpyAsync(O(i), X(i), bytes, hipMemcpyDeviceToDevice, 0)); }, 20, 100), 2.0 * bytes);

    char tag[128];
#define COPY(U, P)                                                                                                     \
    for (int gm : {4, 8, 16}) {                                                                                        \
        snprintf(tag, sizeof tag, "copy U=%d policy=%d grid=256x%d", U, P, gm);                                        \
        report(tag, time_ms([&](int i) { hipLaunchKernelGGL((copy_k<U, P>), dim3(256 * gm), dim3(256), 0, 0, X(i), O(i), \
                                                            (long long)(n4 / (256 * U))); }, 50, 200), 2.0 * bytes);   \
    }
    COPY(1, 0) COPY(2, 0) COPY(4, 0) COPY(8, 0)
    COPY(4, 1) COPY(4, 2) COPY(4, 3) COPY(8, 1)
#define READ(U)                                                                                                        \
    for (int gm : {4, 8}) {                                                                                            \
        snprintf(tag, sizeof tag, "read U=%d grid=256x%d", U, gm);                                                     \
        report(tag, time_ms([&](int i) { hipLaunchKernelGGL((read_k<U>), dim3(256 * gm), dim3(256), 0, 0, X(i), (float*)O(0), \
                                                            (long long)(n4 / (256 * U))); }, 50, 200), 1.0 * bytes);   \
    }
    READ(1) READ(4) READ(8)
    for (int gm : {4, 8}) {
        snprintf(tag, sizeof tag, "write U=4 grid=256x%d", gm);
        report(tag, time_ms([&](int i) { hipLaunchKernelGGL((write_k<4>), dim3(256 * gm), dim3(256), 0, 0, O(i), (long long)(n4 / 1024), 1.f); }, 50, 200), 1.0 * bytes);
    }

    // STFT shape: frames of 512 new samples; as many frames as the buffer holds
    const int nframes = (int)(bytes / 2048) - 2;
    const double moved = (double)nframes * 4100;
#define SHAPE(D, L, RS, P, ST)                                                                                         \
    for (int run : {4, 8, 16, 32}) {                                                                                   \
        const int waves = (nframes + run - 1) / run, blocks = (waves + 3) / 4;                                         \
        snprintf(tag, sizeof tag, "stft-shape d=%d lds=%d row=%d pol=%d store=%d run=%d", D, L, RS, P, ST, run);       \
        report(tag, time_ms([&](int i) { hipLaunchKernelGGL((stft_shape_k<D, L, RS, P, ST>), dim3(blocks), dim3(256), 0, 0, \
                                                            (const float*)X(i), (float*)O(i), run, nframes); }, 50, 200), moved); \
    }
    if (argc > 3) {                                         // quick mode: write-only policies, then the shapes that matter
        for (int gm : {8, 16}) {
            snprintf(tag, sizeof tag, "write U=4 nt grid=256x%d", gm);
            report(tag, time_ms([&](int i) { hipLaunchKernelGGL((copy_k<4, 2>), dim3(256 * gm), dim3(256), 0, 0, X(0), O(i), (long long)(n4 / 1024)); }, 50, 200), 2.0 * bytes);
        }
    }
    SHAPE(1, 36864, 513, 0, 0) SHAPE(1, 36864, 513, 1, 0) SHAPE(1, 36864, 513, 2, 0) SHAPE(1, 36864, 513, 3, 0)
    SHAPE(1, 36864, 513, 0, 1) SHAPE(1, 36864, 513, 1, 1) SHAPE(1, 36864, 513, 2, 1) SHAPE(1, 36864, 513, 3, 1)
    SHAPE(1, 0, 513, 0, 1) SHAPE(1, 0, 513, 1, 1) SHAPE(2, 36864, 513, 1, 1)
    return 0;
}
