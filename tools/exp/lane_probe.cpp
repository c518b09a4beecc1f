// lane_probe.cpp — what does the exact IIR bank's output pass (iir_lane_kernel) cost per sample, and at which clock?
// One lane = one recurrence with its state in registers and wave-uniform coefficients (scalar operands), the decimator's shape:
// ORD = 12, 2 ORD + 1 = 25 float64 multiply-adds per sample, nothing else (no loads inside the loop: samples are synthesised from a
// register).  Reports, for 1 / 2 / 4 wavefronts per SIMD and for zero / noise-like data: shader cycles per sample and wave
// (s_memtime), the clock (s_memtime against the 100 MHz s_memrealtime) and ns per sample — i.e. whether the bank's ~135 ns per
// sample (~300 cycles at 2.2 GHz) is the instruction stream at a lower clock, or something the real kernel adds.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/lane_probe.bin tools/exp/lane_probe.cpp ; run: tools/exp/lane_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ORD = 12;
struct Coef { double b[ORD + 1], a[ORD + 1]; };

__global__ void __launch_bounds__(64) lane(const Coef c, int n, double seed_scale, double* out, long long* clk) {
    double z[ORD];
#pragma unroll
    for (int s = 0; s < ORD; ++s) z[s] = 0.0;
    double x = seed_scale * (double)(threadIdx.x + 1 + 64 * (blockIdx.x % 7)), acc = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; ++k) {
        const double y = __builtin_fma(c.b[0], x, z[0]);
#pragma unroll
        for (int s = 0; s + 1 < ORD; ++s) z[s] = __builtin_fma(-c.a[s + 1], y, __builtin_fma(c.b[s + 1], x, z[s + 1]));
        z[ORD - 1] = __builtin_fma(-c.a[ORD], y, c.b[ORD] * x);
        acc += y;
        x = __builtin_fma(x, -0.999, 1e-3 * y);              // the next "sample": data dependent, bounded, costs two more instructions
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    if (acc == 1.2345e300) out[0] = acc + z[0];
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int simds = p.multiProcessorCount * 4, n = 1024;
    Coef c{};
    for (int i = 0; i <= ORD; ++i) { c.b[i] = 0.01 * (i + 1); c.a[i] = i ? 0.05 / (i + 1) : 1.0; }      // a stable stand-in
    double* out; long long* clk;
    CK(hipMalloc(&out, 8));
    CK(hipMalloc(&clk, sizeof(long long) * 2 * simds * 4));
    for (double scale : {0.0, 1e-3}) {
        for (int wps : {1, 2, 4}) {
            const int waves = simds * wps;
            for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(lane, dim3(waves), dim3(64), 0, 0, c, n, scale, out, clk);      // sustained clocks
            CK(hipDeviceSynchronize());
            std::vector<long long> h(2 * waves);
            CK(hipMemcpy(h.data(), clk, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0;
            for (int w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
            cyc /= waves; rt /= waves;
            printf("data %-5s %d waves/SIMD: %7.1f cycles / sample / wave, clock %.2f GHz, %6.1f ns / sample (27 float64 instructions per sample)\n",
                   scale == 0.0 ? "zero" : "noise", wps, cyc / n, cyc / rt * 0.1, rt * 10.0 / n);
        }
    }
    return 0;
}
