// gcc.hip — K5: generalised cross-correlation with phase transform (GCC-PHAT) and the delay
// read-out, for gfx950.
//
// Reference semantics:
//   generalized_cross_correlation(d0, d1)                     friture/signal/correlation.py:24-43
//       d0 -= mean(d0); d1 -= mean(d1)          (in place on the caller's arrays)
//       w = numpy.hanning(L);  D0 = rfft(d0 w);  D1 = rfft(d1 w)
//       G = conj(D0) D1;  W = 1 / (1e-10 max|G| + |G|);  Xcorr = irfft(W G)
//   read-out (Delay_Estimator_Widget.handle_new_data)          friture/delay_estimator.py:134-176
//       smoothed = 0.3 Xcorr + 0.7 old;  i = argmax |smoothed|;  delay_ms = 1e3 i / rate (wrapped);
//       confidence from |smoothed[i]| / (3 std(smoothed))
//
// Kernel shape: one workgroup of kGccThreads (512) threads per channel pair / window, float64.  A real
// transform of length L is a complex transform of length M = L/2; M = R * M2 is split DIT-wise
// into R interleaved sub-transforms of length M2 <= 6144 so that one sub-transform (96 KB of
// complex doubles) fits in LDS next to nothing else; sub-spectra and the cross spectrum pass
// through an HBM scratch slab owned by the workgroup (L2 resident: 64 bytes per complex point of
// M).  The default L = 24000 = 2 * 2 * 6000 (friture/delay_estimator.py:114-115) runs R = 2; handles of at most 32 pairs (the
// widget's one pair) run it as R = 4 sub-transforms of 3000 points — a pair is then eight forward and four inverse workgroups of the
// launches-of-phases path (gcc_fwd / cross / pack / inv_kernel), three per CU.
#include <cmath>

#include "common.h"
#include "fft_mixed.h"
#include "fft_core.h"
#include "fft_static.h"

namespace frt {

#ifndef FRT_GCC_THREADS          // threads per workgroup of every kernel here.  Measured 256 / 384 / 512 / 768 / 1024 (1024 pairs of the default
#define FRT_GCC_THREADS 512      // window): 0.82 / 0.70 / 0.60 / 0.61 / 0.64 ms — half as many waves at the barriers, twice the points per thread
#endif
constexpr int kGccThreads = FRT_GCC_THREADS;
constexpr int kGccMaxM2 = 6144;
constexpr int kGccMaxB = (kGccMaxM2 / 2 + kGccThreads - 1) / kGccThreads;      // radix-2 butterflies of the largest sub-transform per thread
constexpr int kGccMaxR = 4;

struct GccArgs {
    const double* d0;      // [pairs][L]
    const double* d1;
    double* xcorr;         // [pairs][L]
    int* argmax;           // [pairs] or null
    double* means;         // [pairs][2] or null
    const double* window;  // [L] numpy.hanning(L)
    const double* twm;     // [M] exp(-2 pi i t / M)
    const double* tw2;     // per-pass twiddle tables of the M2-point plan (fft_mixed.h, make_pass_twiddles)
    const double* tws;     // tables of the compile-time plan (fft_static.h) when M2 = 6000
    const double* twl;     // [M+1] exp(-2 pi i k / L)
    const double* dw;      // [M+1] complex: rfft(window) — the mean of a signal leaves its spectrum as mean * dw (gcc_phat_kernel)
    double* psum;          // [pairs][2][R] sums of the samples a forward workgroup loaded (R = 4 with the 3000-point plan), else null
    double* scratch;       // [pairs][4 M + 2] complex
    MixedPlan plan;        // for M2
    int L, M, M2, R;
    int vec;               // d0, d1 and xcorr are 16-byte aligned
    long long* prof;       // FRT_GCC_PROFILE: phase time stamps of workgroup 0 (100 MHz counter), else null
};
#define GCC_STAMP(i)                                                           \
    do {                                                                       \
        if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.prof[i] = wall_clock64(); \
    } while (0)

template <typename T>
__device__ T block_sum(T v, T* red) {
    const int tid = threadIdx.x;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    T s = 0;
    for (int w = 0; w < kGccThreads / 64; ++w) s += red[w];
    return s;
}

__device__ double block_max(double v, double* red) {
    const int tid = threadIdx.x;
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < kGccThreads / 64; ++w) s = fmax(s, red[w]);
    return s;
}

// The sub-transform engine: the compile-time plan 6 x 10 x 10 x 10 (fft_static.h) for the default window's 6000 points,
// the run-time mixed-radix plan for every other 5-smooth length.
// ST: 0 the run-time plan, 1 the 6000-point plan (R = 2 of the default window), 2 the 3000-point plan 3 x 10 x 10 x 10 (R = 4 of the
// default window: small batches as launches of their phases — eight forward and four inverse workgroups of 48 KB per pair, three per CU)
constexpr int kGccStaticM2 = 6000, kGccStaticM2Small = 3000;
constexpr int kGccSlotsMax = (kGccMaxM2 + kGccThreads - 1) / kGccThreads;      // points of a sub-transform per thread: 6
template <int ST>
__device__ __forceinline__ void gcc_fft(cpx<double>* buf, const GccArgs& a, int tid) {
    if constexpr (ST == 1) static_fft_forward<double, kGccThreads, 6, 10, 10, 10>(buf, (const cpx<double>*)a.tws, tid);
    else if constexpr (ST == 2) static_fft_forward<double, kGccThreads, 3, 10, 10, 10>(buf, (const cpx<double>*)a.tws, tid);
    else fft_mixed_forward<double, kGccMaxB>(buf, (const cpx<double>*)a.tw2, a.plan, tid, kGccThreads);
}

// Z_s[kk] = sum_r W_M^{r kk} S[s][r][kk mod M2], kk < M: the last (radix-R, decimation in time) step of the M-point transform
template <int R>
__device__ __forceinline__ cpx<double> gcc_zfull(const cpx<double>* S, const cpx<double>* twm, int s, int kk, int M, int M2) {
    using C = cpx<double>;
    int kp = kk;
#pragma unroll
    for (int q = 1; q < R; ++q) kp = kp >= M2 ? kp - M2 : kp;
    C acc = S[((size_t)s * R) * M2 + kp];
#pragma unroll
    for (int r = 1; r < R; ++r) {
        int idx = r * kk;                            // < R M
#pragma unroll
        for (int q = 1; q < r + 1; ++q) idx = idx >= M ? idx - M : idx;
        acc = acc + cmul(twm[idx], S[((size_t)s * R + r) * M2 + kp]);
    }
    return acc;
}

// D[kk] of the real signal whose even / odd samples rode in Z: A = Z[kk], B = Z[M - kk] (both taken mod M), t = exp(-2 pi i kk / L)
__device__ __forceinline__ cpx<double> gcc_unpack(cpx<double> A, cpx<double> Bz, cpx<double> t) {
    using C = cpx<double>;
    const C B = cconj(Bz);
    const C Sm = A + B, D = A - B;
    const C u = cmul(t, D);
    return {0.5 * (Sm.x + u.y), 0.5 * (Sm.y - u.x)};
}

// Sample pair p of a window (16-byte load when the window is aligned; a view into a ring need not be)
__device__ __forceinline__ double2 gcc_pair_at(const double* sig, int p, int vec) {
    return vec ? ((const double2*)sig)[p] : double2{sig[2 * p], sig[2 * p + 1]};
}

// Sum of a window's L samples over the workgroup's threads, eight independent loads per thread in flight at a time (a
// loop of one load per trip pays the memory latency L / 2048 times over: 12 round trips for the default window).
__device__ __forceinline__ double gcc_thread_sum(const double* sig, int L, int vec, int tid) {
    double acc = 0.0;
    const int np = L / 2;
    for (int base = 0; base < np; base += 8 * kGccThreads) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = base + u * kGccThreads + tid;
            v[u] = p < np ? gcc_pair_at(sig, p, vec) : double2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
    }
    return acc;
}

// (x - mean) w of sub-transform r into the LDS array, every thread's (at most six) loads in flight together; returns the sum of the
// thread's samples x
template <int R>
__device__ __forceinline__ double gcc_load_sub(const GccArgs& a, const double* sig, double mean, int r, cpx<double>* buf, int tid) {
    const double2* wn = (const double2*)a.window;
    const int M2 = a.M2;
    double2 x[kGccSlotsMax], w[kGccSlotsMax];
#pragma unroll
    for (int i = 0; i < kGccSlotsMax; ++i) {
        const int m = tid + i * kGccThreads;
        x[i] = w[i] = double2{0.0, 0.0};
        if (m < M2) {
            x[i] = gcc_pair_at(sig, R * m + r, a.vec);
            w[i] = wn[R * m + r];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < kGccSlotsMax; ++i) {
        const int m = tid + i * kGccThreads;
        acc += x[i].x + x[i].y;                             // (slots beyond M2 hold zeros)
        if (m < M2) buf[m] = {(x[i].x - mean) * w[i].x, (x[i].y - mean) * w[i].y};
    }
    return acc;
}

// Input of inverse sub-transform r into the LDS array: conj( W_M^{-r k} sum_q W_R^{-r q} Zi[k + q M2] ) (the conjugate turns
// the forward engine into the inverse: ifft(u) = conj(fft(conj u)) / n), a thread's loads in flight together
template <int R>
__device__ __forceinline__ void gcc_load_inverse(const cpx<double>* Zi, const cpx<double>* twm, int r, int M2, cpx<double>* buf, int tid) {
    using C = cpx<double>;
    constexpr int G = R <= 2 ? kGccSlotsMax : 2;           // slots per group: G R + G loads in flight
#pragma unroll
    for (int i0 = 0; i0 < kGccSlotsMax; i0 += G) {
        C z[G][R], tw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = tid + (i0 + g) * kGccThreads;
            tw[g] = {1.0, 0.0};
#pragma unroll
            for (int q = 0; q < R; ++q) z[g][q] = {0.0, 0.0};
            if (k < M2) {
#pragma unroll
                for (int q = 0; q < R; ++q) z[g][q] = Zi[k + q * M2];
                tw[g] = twm[r * k];                          // r k < M
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = tid + (i0 + g) * kGccThreads;
            if (k < M2) {
                C acc = z[g][0];
#pragma unroll
                for (int q = 1; q < R; ++q) acc = acc + cmul(cconj(twm[((r * q) % R) * M2]), z[g][q]);      // W_R^{-r q} = conj(W_M^{(r q mod R) M2})
                buf[k] = cconj(cmul(cconj(tw[g]), acc));
            }
        }
    }
}

// Forward sub-transforms of ONE signal by one workgroup, R <= 2.  The signal is read once: thread t takes the sample pairs
// p = R (t + 1024 i) + r for every r (R consecutive 16-byte words per slot), keeps them in registers while the block sum
// gives the mean, turns them into (x - mean) w in place, and feeds sub-transform after sub-transform from registers — no
// second pass over the window for the mean, no re-read per sub-transform.  Returns the mean.
constexpr int kGccSlots = (kGccMaxM2 + kGccThreads - 1) / kGccThreads;      // 6
template <int R, int ST>
__device__ __forceinline__ double gcc_forward_signal(const GccArgs& a, const double* sig, cpx<double>* Sdst, cpx<double>* buf,
                                                    double* red, int tid) {
    static_assert(R <= 2, "register budget");
    const int M2 = a.M2;
    double2 x[kGccSlots][R], w[kGccSlots][R];
    const double2* wn = (const double2*)a.window;
#pragma unroll
    for (int i = 0; i < kGccSlots; ++i) {
        const int m = tid + i * kGccThreads;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[i][r] = w[i][r] = double2{0.0, 0.0};
            if (m < M2) {
                const int p = R * m + r;
                x[i][r] = a.vec ? ((const double2*)sig)[p] : double2{sig[2 * p], sig[2 * p + 1]};
                w[i][r] = wn[p];
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < kGccSlots; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) acc += x[i][r].x + x[i][r].y;
    const double mean = block_sum(acc, red) / (double)a.L;
#pragma unroll
    for (int i = 0; i < kGccSlots; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) x[i][r] = double2{(x[i][r].x - mean) * w[i][r].x, (x[i][r].y - mean) * w[i][r].y};
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i = 0; i < kGccSlots; ++i) {
            const int m = tid + i * kGccThreads;
            if (m < M2) buf[m] = {x[i][r].x, x[i][r].y};
        }
        __syncthreads();
        gcc_fft<ST>(buf, a, tid);
        cpx<double>* dst = Sdst + (size_t)r * M2;
        for (int k = tid; k < M2; k += kGccThreads) dst[k] = buf[k];
        __syncthreads();
    }
    return mean;
}

// dw[k] = rfft(window)[k], k <= M = L / 2, by the definition (compensated sums; the twiddles are the table's own entries:
// exp(-2 pi i j / L) = twl[j] for j <= M, conj(twl[L - j]) above).  Runs once per handle.
__global__ void __launch_bounds__(256) gcc_window_rfft_kernel(const double* __restrict__ window, const double* __restrict__ twl, int L, int M,
                                                              double* __restrict__ dw) {
    using C = cpx<double>;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > M) return;
    const C* tw = (const C*)twl;
    double sr = 0.0, si = 0.0, cr = 0.0, ci = 0.0;
    int j = 0;                                           // n k mod L
    for (int n = 0; n < L; ++n) {
        const C e = j <= M ? tw[j] : cconj(tw[L - j]);
        const double w = window[n];
        const double yr = w * e.x - cr, tr = sr + yr;
        cr = (tr - sr) - yr;
        sr = tr;
        const double yi = w * e.y - ci, ti = si + yi;
        ci = (ti - si) - yi;
        si = ti;
        j += k;
        if (j >= L) j -= L;
    }
    dw[2 * k] = sr;
    dw[2 * k + 1] = si;
}

}  // namespace frt
#include "gcc_resident.h"
constexpr int kResThreads = 512;      // (768 threads — the transform in 74 registers — was measured 4 % slower: 23 spilled, profiles/r06_gcc_batch.txt)
namespace frt {

// One window pair per workgroup, everything between the two signals and the correlation in this launch.  What passes
// through the pair's scratch slab is only what cannot stay on the CU: of the 2 R sub-spectra (M2 complex each, one LDS
// array's worth) all but the last, which the cross spectrum reads where the transform left it.  The cross spectrum itself
// never leaves the registers: thread t owns the bin pairs (k, M - k), k = t + 1024 i — the real-transform unpack needs
// exactly that pair of Z, and so does the Hermitian packing in front of the inverse transform — and only the block maximum
// of |G| (the PHAT weight's regulariser) is exchanged in between.  For R <= 2 the packed inverse input goes from those
// registers to the inverse sub-transforms through LDS: R = 1 writes it in place; R = 2 needs Zi[k] +- Zi[k + M2], and
// Zi[k + M2] = Zi[M - (M2 - k)] sits with the owner of the mirrored pair, so the upper halves are exchanged through the
// LDS array, the r = 0 input is formed in it and the r = 1 input waits in registers.
// (R = 4, up to 13 bin pairs per thread, computes the cross spectrum twice — once for the maximum, once to pack — and
// passes the packed input through the slab.)
template <int R>
struct GccSub {
    const cpx<double>* p[2][R];             // sub-spectrum (s, r): the scratch slab, or LDS for the last one
};

template <int R>
__device__ __forceinline__ cpx<double> gcc_zfull_at(const GccSub<R>& sub, const cpx<double>* twm, int s, int kk, int M, int M2) {
    using C = cpx<double>;
    int kp = kk;
#pragma unroll
    for (int q = 1; q < R; ++q) kp = kp >= M2 ? kp - M2 : kp;
    C acc = sub.p[s][0][kp];
#pragma unroll
    for (int r = 1; r < R; ++r) {
        int idx = r * kk;                            // < R M
#pragma unroll
        for (int q = 1; q < r + 1; ++q) idx = idx >= M ? idx - M : idx;
        acc = acc + cmul(twm[idx], sub.p[s][r][kp]);
    }
    return acc;
}

template <int R, int ST>
__global__ void __launch_bounds__(kGccThreads) gcc_phat_kernel(const GccArgs a) {
    using C = cpx<double>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* buf = (C*)smem;                                  // M2 points
    double* red = (double*)(buf + a.M2);                // 16 doubles + 16 ints
    int* redi = (int*)(red + 16);

    const int tid = threadIdx.x;
    const int pair = blockIdx.x;
    const int L = a.L, M = a.M, M2 = a.M2;
    // (no array that the loops over s and r index at run time: such an array lives in scratch memory, every use is a scratch load and
    // a wait for EVERY load in flight — the signal pointers, the means and the sub-spectrum pointers were 80 bytes of it)
    const double* const sig0 = a.d0 + (size_t)pair * L;
    const double* const sig1 = a.d1 + (size_t)pair * L;
    C* S = (C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2);   // [2][R][M2] sub-spectra
    C* Zi = S + 2 * (size_t)M + (M + 1);                         // [M] packed inverse input (R = 4 only)
    const C* twm = (const C*)a.twm;
    const C* twl = (const C*)a.twl;
    const C* dwt = (const C*)a.dw;

    GCC_STAMP(0);
    GCC_STAMP(1);                                        // (the means have no phase of their own any more)
    // ---- forward sub-transforms: S[s][r][k'] = FFT_M2( z_s[R m + r] ); the last one stays in LDS ---------------
    // The mean is NOT subtracted here: (x - mean) w has the spectrum rfft(x w) - mean rfft(w), so the samples' sum is taken while
    // they pass through on their way into the sub-transforms (every sample pair is loaded exactly once over r) and the
    // correction mean * dw[k] is applied where the cross spectrum forms D_s[k].  The separate pass over the two windows that
    // computed the means first was a fifth of this kernel at a full chip (28 of 150 us per pair: the chip's whole HBM stream).
    // (Requesting the samples of the next sub-transform while the one before it runs — twelve 16-byte loads per thread held across
    // it — was measured: 0.58 against 0.54 ms for 1024 pairs; the kernel sits at its 256 registers and spills.)
    double mean0 = 0.0, mean1 = 0.0;
    for (int s = 0; s < 2; ++s) {
        double acc = 0.0;
        for (int r = 0; r < R; ++r) {
            // (every load of the sub-transform's samples and window values in flight together: as a loop of one pair of loads per
            // trip — not unrolled by the compiler — each of the twelve trips waited for its own round trip: 8-12 us per sub-transform)
            acc += gcc_load_sub<R>(a, s == 0 ? sig0 : sig1, 0.0, r, buf, tid);
            __syncthreads();
            if (s == 0 && r == 0) GCC_STAMP(2);
            gcc_fft<ST>(buf, a, tid);
            if (s == 0 && r == 0) GCC_STAMP(3);
            const bool last = s == 1 && r == R - 1;
            C* dst = S + ((size_t)s * R + r) * M2;
            if (!last) {
                for (int k = tid; k < M2; k += kGccThreads) dst[k] = buf[k];
                __syncthreads();
            }
        }
        const double m = block_sum(acc, red) / (double)L;
        if (s == 0) mean0 = m;
        else mean1 = m;
    }
    if (tid == 0 && a.means) {
        a.means[2 * pair] = mean0;
        a.means[2 * pair + 1] = mean1;
    }
    const double mean[2] = {mean0, mean1};                      // (indexed by unrolled loops only)
    GccSub<R> sub;                                               // sub-spectrum (s, r): the scratch slab, or LDS for the last one
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r)
            sub.p[s][r] = (s == 1 && r == R - 1) ? (const C*)buf : (const C*)(S + ((size_t)s * R + r) * M2);
    __threadfence_block();
    __syncthreads();
    GCC_STAMP(4);

    // ---- cross spectrum of the bin pairs (k, M - k) and its maximum magnitude -----------------------------------
    constexpr bool KEEP = R <= 2;                        // M <= 12288: at most 7 pairs per thread stay in registers
    constexpr int NP = KEEP ? (2 * kGccMaxM2 / 2 + 1 + kGccThreads - 1) / kGccThreads : 1;
    C g_lo[NP], g_hi[NP];
    auto cross = [&](int k, C& glo, C& ghi) {
        const int km = k == 0 ? 0 : M - k;                  // Z index of the partner bin
        const C tk = twl[k], tm = twl[M - k];
        C d_lo[2], d_hi[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const C Za = gcc_zfull_at<R>(sub, twm, s, k, M, M2), Zb = gcc_zfull_at<R>(sub, twm, s, km, M, M2);
            const C wl = dwt[k], wh = dwt[M - k];
            d_lo[s] = gcc_unpack(Za, Zb, tk);               // D_s[k] of x w ...
            d_hi[s] = gcc_unpack(Zb, Za, tm);               // D_s[M - k]
            d_lo[s] = {d_lo[s].x - mean[s] * wl.x, d_lo[s].y - mean[s] * wl.y};      // ... of (x - mean) w
            d_hi[s] = {d_hi[s].x - mean[s] * wh.x, d_hi[s].y - mean[s] * wh.y};
        }
        glo = cmul(cconj(d_lo[0]), d_lo[1]);
        ghi = cmul(cconj(d_hi[0]), d_hi[1]);
    };
    double gmax = 0.0;
    if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = tid + i * kGccThreads;
            g_lo[i] = g_hi[i] = {0.0, 0.0};
            if (2 * k <= M) cross(k, g_lo[i], g_hi[i]);
            gmax = fmax(gmax, fmax(sqrt(g_lo[i].x * g_lo[i].x + g_lo[i].y * g_lo[i].y), sqrt(g_hi[i].x * g_hi[i].x + g_hi[i].y * g_hi[i].y)));
            asm volatile("" ::: "memory");                   // one pair's loads in flight at a time (register budget; the phase is
                                                             // bound by its float64 arithmetic — two in flight measured equal)
        }
    } else {
        for (int k = tid; 2 * k <= M; k += kGccThreads) {
            C glo, ghi;
            cross(k, glo, ghi);
            gmax = fmax(gmax, fmax(sqrt(glo.x * glo.x + glo.y * glo.y), sqrt(ghi.x * ghi.x + ghi.y * ghi.y)));
        }
    }
    gmax = block_max(gmax, red);                             // (its barriers: every read of the LDS sub-spectrum is done)
    GCC_STAMP(5);

    // ---- PHAT weighting and packing for the inverse real transform ---------------------------------------
    // (zk, zm) = Zi[k], Zi[M - k] from the pair's cross-spectrum values
    auto pack = [&](int k, C A, C B, C& zk, C& zm) {
        const double wa = 1.0 / (1e-10 * gmax + sqrt(A.x * A.x + A.y * A.y));
        const double wb = 1.0 / (1e-10 * gmax + sqrt(B.x * B.x + B.y * B.y));
        A = {A.x * wa, A.y * wa};
        B = {B.x * wb, B.y * wb};
        if (k == 0) A.y = B.y = 0.0;                        // irfft ignores the imaginary part of the edge bins 0 and M
        const C tk = cconj(twl[k]), tm = cconj(twl[M - k]);
        {
            const C Bc = cconj(B), Sm = A + Bc, D = A - Bc, u = cmul(tk, D);
            zk = {0.5 * (Sm.x - u.y), 0.5 * (Sm.y + u.x)};
        }
        {
            const C Ac = cconj(A), Sm = B + Ac, D = B - Ac, u = cmul(tm, D);
            zm = {0.5 * (Sm.x - u.y), 0.5 * (Sm.y + u.x)};
        }
    };
    C x1[KEEP && R == 2 ? NP : 1];                            // R = 2: the r = 1 input of the thread's slots, conjugated
    if constexpr (R == 1) {
        // the one inverse transform's input, conj(Zi[k]), straight into the LDS array
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = tid + i * kGccThreads;
            if (2 * k <= M) {
                C zk, zm;
                pack(k, g_lo[i], g_hi[i], zk, zm);
                buf[k] = cconj(zk);
                if (k > 0) buf[M - k] = cconj(zm);
            }
        }
        __syncthreads();
    } else if constexpr (R == 2) {
        // Zi[k] (k < M2) stays with its owner, Zi[M - k] = upper half element M2 - k goes through LDS to the owner of that slot
        C zlo[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = tid + i * kGccThreads;
            zlo[i] = {0.0, 0.0};
            if (2 * k <= M) {
                C zk, zm;
                pack(k, g_lo[i], g_hi[i], zk, zm);
                if (k < M2) zlo[i] = zk;
                if (k > 0) buf[M2 - k] = zm;                 // k = M2: zm = zk = Zi[M2], upper element 0
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = tid + i * kGccThreads;
            x1[i] = {0.0, 0.0};
            if (k < M2) {
                const C hi = buf[k];                         // Zi[k + M2]
                const C s0 = zlo[i] + hi, s1 = cmul(cconj(twm[k]), zlo[i] - hi);
                zlo[i] = cconj(s0);
                x1[i] = cconj(s1);
            }
        }
        __syncthreads();                                     // every upper-half element has been read
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = tid + i * kGccThreads;
            if (k < M2) buf[k] = zlo[i];
        }
        __syncthreads();
    } else {
        for (int k = tid; 2 * k <= M; k += kGccThreads) {
            C glo, ghi, zk, zm;
            cross(k, glo, ghi);
            pack(k, glo, ghi, zk, zm);
            Zi[k] = zk;
            if (k > 0) Zi[M - k] = zm;
        }
        __threadfence_block();
        __syncthreads();
    }

    GCC_STAMP(6);
    // ---- inverse sub-transforms: z[R m + r] = (1/M) IFFT_M2( W_M^{-r k'} sum_q W_R^{-r q} Zi[k' + q M2] ) ---
    double* out = a.xcorr + (size_t)pair * L;
    const double inv = 1.0 / (double)M;
    double best = -1.0;
    int besti = 0;
    for (int r = 0; r < R; ++r) {
        if constexpr (R == 2) {
            if (r == 1) {
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int k = tid + i * kGccThreads;
                    if (k < M2) buf[k] = x1[i];
                }
                __syncthreads();
            }
        } else if constexpr (R > 2) {
            gcc_load_inverse<R>(Zi, twm, r, M2, buf, tid);
            __syncthreads();
        }
        gcc_fft<ST>(buf, a, tid);
        for (int m = tid; m < M2; m += kGccThreads) {
            const int t = 2 * (R * m + r);
            const double re = buf[m].x * inv, im = -buf[m].y * inv;
            if (a.vec) {
                *(double2*)(out + t) = double2{re, im};
            } else {
                out[t] = re;
                out[t + 1] = im;
            }
            if (fabs(re) > best || (fabs(re) == best && t < besti)) { best = fabs(re); besti = t; }
            if (fabs(im) > best || (fabs(im) == best && t + 1 < besti)) { best = fabs(im); besti = t + 1; }
        }
        __syncthreads();
    }

    GCC_STAMP(7);
    // ---- argmax |xcorr| (first index on ties, as numpy.argmax) ---------------------------------------------
    if (a.argmax) {
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_down(best, o, 64);
            const int oi = __shfl_down(besti, o, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kGccThreads / 64; ++w)
                if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
            a.argmax[pair] = besti;
        }
    }
    GCC_STAMP(8);
}

// Smoothing + statistics of the read-out: sm = alpha x + (1 - alpha) old (or x when old is null);
// per pair: argmax |sm|, sm[argmax], std(sm) (two-pass, as numpy.std).
__global__ void __launch_bounds__(kGccThreads) gcc_readout_kernel(const double* __restrict__ x, const double* __restrict__ old,
                                                                  double* __restrict__ sm, int L, double alpha,
                                                                  int* __restrict__ argmax, double* __restrict__ stats) {
    __shared__ double red[16];
    __shared__ int redi[16];
    const int tid = threadIdx.x, pair = blockIdx.x;
    const double* xp = x + (size_t)pair * L;
    const double* op = old ? old + (size_t)pair * L : nullptr;
    double* sp = sm + (size_t)pair * L;
    double best = -1.0, acc = 0.0;
    int besti = 0;
    for (int t = tid; t < L; t += kGccThreads) {
        const double v = op ? alpha * xp[t] + (1.0 - alpha) * op[t] : xp[t];
        sp[t] = v;
        acc += v;
        if (fabs(v) > best) { best = fabs(v); besti = t; }
    }
    const double mean = block_sum(acc, red) / (double)L;
    double var = 0.0;
    for (int t = tid; t < L; t += kGccThreads) {
        const double d = sp[t] - mean;
        var += d * d;
    }
    var = block_sum(var, red) / (double)L;
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kGccThreads / 64; ++w)
            if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
        argmax[pair] = besti;
        stats[2 * pair] = sp[besti];
        stats[2 * pair + 1] = sqrt(var);
    }
}


// ---- the same transform as three phases of SMALL workgroups ---------------------------------------------------------
// gcc_phat_kernel keeps a window pair in one workgroup: a batch of 100 pairs (BASELINE configs[4]) then
// occupies 100 of the 256 CUs.  For batches that do not fill the chip the work of a pair is dealt to more workgroups —
// 2 R forward sub-transforms, the cross spectrum in slices, R inverse sub-transforms — at the price of four kernel
// boundaries; every phase reads what the previous one left in the pair's scratch slab (same layout as above).
template <int R, int ST>
__global__ void __launch_bounds__(kGccThreads) gcc_fwd_kernel(const GccArgs a, unsigned long long* __restrict__ gmax) {
    using C = cpx<double>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* buf = (C*)smem;
    double* red = (double*)(buf + a.M2);
    const int tid = threadIdx.x, pair = blockIdx.y;
    const int L = a.L, M = a.M, M2 = a.M2;
    C* S = (C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2);
    if constexpr (R <= 2) {
        // one workgroup per signal: the signal is read once and feeds all its sub-transforms from registers
        const int s = blockIdx.x;
        const double* sig = (s ? a.d1 : a.d0) + (size_t)pair * L;
        const double mean = gcc_forward_signal<R, ST>(a, sig, S + (size_t)s * R * M2, buf, red, tid);
        if (tid == 0) {
            if (a.means) a.means[2 * pair + s] = mean;
            if (s == 0) gmax[pair] = 0ull;
        }
    } else {
        const int s = blockIdx.x / R, r = blockIdx.x - s * R;
        const double* sig = (s ? a.d1 : a.d0) + (size_t)pair * L;
        if constexpr (ST == 2) {
            // every sample is read once over the signal's R workgroups: the transform of x w, and the sum of this workgroup's samples
            // for the mean, which gcc_cross_kernel removes in the spectrum (mean * rfft(window), as gcc_phat_kernel does)
            const double part = block_sum(gcc_load_sub<R>(a, sig, 0.0, r, buf, tid), red);
            if (tid == 0) {
                a.psum[((size_t)pair * 2 + s) * R + r] = part;
                if (s == 0 && r == 0) gmax[pair] = 0ull;
            }
        } else {
            const double mean = block_sum(gcc_thread_sum(sig, L, a.vec, tid), red) / (double)L;
            if (tid == 0 && r == 0) {
                if (a.means) a.means[2 * pair + s] = mean;
                if (s == 0) gmax[pair] = 0ull;
            }
            gcc_load_sub<R>(a, sig, mean, r, buf, tid);
        }
        __syncthreads();
        gcc_fft<ST>(buf, a, tid);
        for (int k = tid; k < M2; k += kGccThreads) S[((size_t)s * R + r) * M2 + k] = buf[k];
    }
}

// DW: the means were not removed from the samples (gcc_fwd_kernel with the 3000-point plan left the sums of its workgroups' samples):
// D_s[k] -= mean_s rfft(window)[k] here
template <int R, bool DW = false>
__global__ void __launch_bounds__(256) gcc_cross_kernel(const GccArgs a, unsigned long long* __restrict__ gmax) {
    using C = cpx<double>;
    __shared__ double red[4];
    const int pair = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const int M = a.M, M2 = a.M2;
    double mean[2] = {0.0, 0.0};
    if constexpr (DW) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) acc += a.psum[((size_t)pair * 2 + s) * R + r];
            mean[s] = acc / (double)a.L;
        }
        if (k == 0 && a.means) {
            a.means[2 * pair] = mean[0];
            a.means[2 * pair + 1] = mean[1];
        }
    }
    const C* S = (const C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2);
    C* G = (C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2) + 2 * (size_t)M;
    const C* twm = (const C*)a.twm;
    const C* twl = (const C*)a.twl;
    double mag = 0.0;
    if (2 * k <= M) {                                    // the bin pair (k, M - k): both need Z[k] and Z[M - k]
        const int km = k == 0 ? 0 : M - k;
        const C tk = twl[k], tm = twl[M - k];
        C d_lo[2], d_hi[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const C Za = gcc_zfull<R>(S, twm, s, k, M, M2), Zb = gcc_zfull<R>(S, twm, s, km, M, M2);
            d_lo[s] = gcc_unpack(Za, Zb, tk);
            d_hi[s] = gcc_unpack(Zb, Za, tm);
            if constexpr (DW) {
                const C wl = ((const C*)a.dw)[k], wh = ((const C*)a.dw)[M - k];
                d_lo[s] = {d_lo[s].x - mean[s] * wl.x, d_lo[s].y - mean[s] * wl.y};
                d_hi[s] = {d_hi[s].x - mean[s] * wh.x, d_hi[s].y - mean[s] * wh.y};
            }
        }
        const C glo = cmul(cconj(d_lo[0]), d_lo[1]), ghi = cmul(cconj(d_hi[0]), d_hi[1]);
        G[k] = glo;
        G[M - k] = ghi;
        mag = fmax(sqrt(glo.x * glo.x + glo.y * glo.y), sqrt(ghi.x * ghi.x + ghi.y * ghi.y));
    }
    for (int o = 32; o > 0; o >>= 1) mag = fmax(mag, __shfl_down(mag, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mag;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(gmax + pair, (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
}

// PHAT weighting + Hermitian packing for the inverse real transform, one thread per bin pair (k, M - k): two weights, two
// packed values (each weight serves both).
__global__ void __launch_bounds__(256) gcc_pack_kernel(const GccArgs a, const unsigned long long* __restrict__ gmax) {
    using C = cpx<double>;
    const int pair = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    const int M = a.M;
    if (2 * k > M) return;
    const C* G = (const C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2) + 2 * (size_t)M;
    C* Zi = (C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2) + 2 * (size_t)M + (M + 1);
    const C* twl = (const C*)a.twl;
    const double gm = __longlong_as_double((long long)gmax[pair]);
    C A = G[k], B = G[M - k];
    const double wa = 1.0 / (1e-10 * gm + sqrt(A.x * A.x + A.y * A.y));
    const double wb = 1.0 / (1e-10 * gm + sqrt(B.x * B.x + B.y * B.y));
    A = {A.x * wa, A.y * wa};
    B = {B.x * wb, B.y * wb};
    if (k == 0) A.y = B.y = 0.0;                        // irfft ignores the imaginary part of the edge bins 0 and M
    const C tk = cconj(twl[k]), tm = cconj(twl[M - k]);
    {
        const C Bc = cconj(B), Sm = A + Bc, D = A - Bc, u = cmul(tk, D);
        Zi[k] = {0.5 * (Sm.x - u.y), 0.5 * (Sm.y + u.x)};
    }
    if (k > 0) {
        const C Ac = cconj(A), Sm = B + Ac, D = B - Ac, u = cmul(tm, D);
        Zi[M - k] = {0.5 * (Sm.x - u.y), 0.5 * (Sm.y + u.x)};
    }
}

// Inverse sub-transform r of a pair; the workgroup leaves the (|value|, first index) of its slice's extremum for
// gcc_argmax_combine_kernel.  (Folding the R slices in the pair's last workgroup to finish was measured: the device-scope
// fences that hand-over needs on a multi-XCD part cost 10 us per launch, twice the tiny launch below.)
template <int R, int ST>
__global__ void __launch_bounds__(kGccThreads) gcc_inv_kernel(const GccArgs a, double* __restrict__ part_val, int* __restrict__ part_idx) {
    using C = cpx<double>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* buf = (C*)smem;
    double* red = (double*)(buf + a.M2);
    int* redi = (int*)(red + 16);
    const int tid = threadIdx.x, pair = blockIdx.y, r = blockIdx.x;
    const int L = a.L, M = a.M, M2 = a.M2;
    const C* Zi = (const C*)a.scratch + (size_t)pair * (4 * (size_t)M + 2) + 2 * (size_t)M + (M + 1);
    const C* twm = (const C*)a.twm;
    gcc_load_inverse<R>(Zi, twm, r, M2, buf, tid);
    __syncthreads();
    gcc_fft<ST>(buf, a, tid);
    double* out = a.xcorr + (size_t)pair * L;
    const double inv = 1.0 / (double)M;
    double best = -1.0;
    int besti = 0;
    for (int m = tid; m < M2; m += kGccThreads) {
        const int t = 2 * (R * m + r);
        const double re = buf[m].x * inv, im = -buf[m].y * inv;
        if (a.vec) {
            *(double2*)(out + t) = double2{re, im};
        } else {
            out[t] = re;
            out[t + 1] = im;
        }
        if (fabs(re) > best || (fabs(re) == best && t < besti)) { best = fabs(re); besti = t; }
        if (fabs(im) > best || (fabs(im) == best && t + 1 < besti)) { best = fabs(im); besti = t + 1; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kGccThreads / 64; ++w)
            if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
        part_val[(size_t)pair * R + r] = best;
        part_idx[(size_t)pair * R + r] = besti;
    }
}


// argmax |xcorr| of a pair from its R slices' extrema (first index on ties, as numpy.argmax)
__global__ void __launch_bounds__(256) gcc_argmax_combine_kernel(const double* __restrict__ part_val, const int* __restrict__ part_idx, int R,
                                                                 int n_pairs, int* __restrict__ argmax) {
    const int pair = blockIdx.x * 256 + threadIdx.x;
    if (pair >= n_pairs) return;
    double best = part_val[(size_t)pair * R];
    int besti = part_idx[(size_t)pair * R];
    for (int r = 1; r < R; ++r) {
        const double v = part_val[(size_t)pair * R + r];
        const int i = part_idx[(size_t)pair * R + r];
        if (v > best || (v == best && i < besti)) { best = v; besti = i; }
    }
    argmax[pair] = besti;
}


// ---- any window length: chirp-z (Bluestein) on a four-step power-of-two transform ----------------------------------
// The delay-range spin box runs from 0.1 s to 1000 s in steps of 0.1 s (delay_estimator.py:222-226): windows of
// L = 2400 r samples, r = 1..10000 — most of them neither 5-smooth nor small enough for the one-workgroup kernel above
// (numpy's rfft takes any length).  Those windows go through the textbook identity
//     X[k] = c[k] sum_n (x[n] c[n]) conj(c)[k - n],   c[n] = exp(-i pi n^2 / L),
// a length-L DFT as a circular convolution of length P = 2^p >= 2L - 1.  Both signals ride in ONE complex transform
// (z = d0 w + i d1 w, separated by conjugate symmetry), the PHAT-weighted cross spectrum is extended to its Hermitian
// full length and goes back through the same machinery: xcorr = Re DFT(conj Y) / L.
// The length-P transforms are four-step: P = R C (both 2^8 .. 2^13), index n = r C + c;
//     forward:  column transforms over r  ->  x W_P^{k_r c}  ->  row transforms over c, result X[k_c R + k_r] AT [k_r][k_c]
//     inverse:  row inverse over k_c  ->  x W_P^{-k_r c}  ->  column inverse over k_r, natural order
// so the product with the chirp spectrum (same transposed layout) and the inverse row transform run in the kernel that
// did the forward row transform, and nothing is ever transposed in memory: three launches per length-L DFT.  The
// length-R / length-C transforms are the STFT's radix-8 engine (fft_core.h), one workgroup per column / row, float64.
struct AnyArgs {
    // geometry
    int L, log2r, log2c;                 // P = R C
    long long P;
    // inputs of the signal-building load (src_mode 1) / of the spectrum load (src_mode 2)
    const double* d0;                    // [pairs][L]
    const double* d1;
    const double* means;                 // [pairs][2]
    const double* window;                // [L]
    const double* chirp;                 // [L] complex: exp(-i pi n^2 / L)
    const double* spec;                  // [pairs][L] complex (src_mode 2: a[n] = spec[n] chirp[n])
    const double* plain;                 // [P] complex (src_mode 0)
    double* work;                        // [pairs][P] complex, the four-step array
    const double* bhat;                  // [P] complex: transform of the chirp filter, transposed layout
    const double* twr;                   // [R] exp(-2 pi i t / R)
    const double* twc;                   // [C]
    // outputs of the final column pass
    double* spec_out;                    // dst_mode 0: [pairs][L] complex X[n] = conv[n] chirp[n]
    double* xcorr;                       // dst_mode 1: [pairs][L] real: Re(conv[n] chirp[n]) / L
};

__device__ __forceinline__ cpx<double> unit_root(long long num, long long den_pow2) {      // exp(-2 pi i num / den)
    double s, c;
    sincospi(-2.0 * ((double)num / (double)den_pow2), &s, &c);
    return {c, s};
}

// column pass of the forward transform: A[k_r C + c] = W_P^{k_r c} sum_r src(r C + c) W_R^{r k_r}
template <int LOG2R, int SRC>
__global__ void __launch_bounds__(Pow2Plan<LOG2R>::TPF) any_col_fwd_kernel(const AnyArgs a) {
    using C = cpx<double>;
    using PL = Pow2Plan<LOG2R>;
    constexpr int R = PL::M, TPF = PL::TPF;
    __shared__ C buf[lds_padded_size(R)];
    const int i = threadIdx.x, col = blockIdx.x, pair = blockIdx.y;
    const long long Cn = 1ll << a.log2c;
    const C* chirp = (const C*)a.chirp;
    double m0 = 0.0, m1 = 0.0;
    if (SRC == 1) { m0 = a.means[2 * pair]; m1 = a.means[2 * pair + 1]; }
    C v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long n = (long long)(i + j * TPF) * Cn + col;
        C x = {0.0, 0.0};
        if (SRC == 0) {
            x = ((const C*)a.plain)[n];
        } else if (n < a.L) {
            if (SRC == 1) {
                const double w = a.window[n];
                x = {(a.d0[(size_t)pair * a.L + n] - m0) * w, (a.d1[(size_t)pair * a.L + n] - m1) * w};
            } else {
                x = ((const C*)a.spec)[(size_t)pair * a.L + n];
            }
            x = cmul(x, chirp[n]);
        }
        v[j] = x;
    }
    const TwTable<double, LOG2R> tw{(const C*)a.twr, 0};
    fft_pow2_forward<double, LOG2R, false>(v, buf, i, tw);
    C* A = (C*)a.work + (size_t)pair * a.P;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long kr = i + j * TPF;
        A[kr * Cn + col] = cmul(v[j], unit_root(kr * col, a.P));
    }
}

// row pass: forward over c; MODE 0 stores the spectrum (transposed layout); MODE 1 multiplies by bhat, inverts over k_c,
// applies W_P^{-k_r c} and stores in place: the array is then ready for the inverse column pass
template <int LOG2C, int MODE>
__global__ void __launch_bounds__(Pow2Plan<LOG2C>::TPF) any_row_kernel(const AnyArgs a) {
    using C = cpx<double>;
    using PL = Pow2Plan<LOG2C>;
    constexpr int Cn = PL::M, TPF = PL::TPF;
    __shared__ C buf[lds_padded_size(Cn)];
    const int i = threadIdx.x, pair = blockIdx.y;
    const long long kr = blockIdx.x;
    C* row = (C*)a.work + (size_t)pair * a.P + kr * Cn;
    C v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = row[i + j * TPF];
    const TwTable<double, LOG2C> tw{(const C*)a.twc, 0};
    fft_pow2_forward<double, LOG2C, false>(v, buf, i, tw);
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) row[i + j * TPF] = v[j];
        return;
    }
    const C* bh = (const C*)a.bhat + kr * Cn;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = cconj(cmul(v[j], bh[i + j * TPF]));       // conj trick for the inverse
    __syncthreads();
    fft_pow2_forward<double, LOG2C, false>(v, buf, i, tw);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long c = i + j * TPF;
        row[c] = cmul(cconj(v[j]), cconj(unit_root(kr * c, a.P)));                // (1/C deferred to the last pass)
    }
}

// inverse column pass and the chirp on the way out
template <int LOG2R, int DST>
__global__ void __launch_bounds__(Pow2Plan<LOG2R>::TPF) any_col_inv_kernel(const AnyArgs a) {
    using C = cpx<double>;
    using PL = Pow2Plan<LOG2R>;
    constexpr int R = PL::M, TPF = PL::TPF;
    __shared__ C buf[lds_padded_size(R)];
    const int i = threadIdx.x, col = blockIdx.x, pair = blockIdx.y;
    const long long Cn = 1ll << a.log2c;
    if (col >= a.L) return;                                   // rows r >= 1 of such a column lie beyond L as well
    const C* A = (const C*)a.work + (size_t)pair * a.P;
    C v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = cconj(A[(long long)(i + j * TPF) * Cn + col]);
    const TwTable<double, LOG2R> tw{(const C*)a.twr, 0};
    fft_pow2_forward<double, LOG2R, false>(v, buf, i, tw);
    const double invp = 1.0 / (double)a.P;
    const C* chirp = (const C*)a.chirp;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long n = (long long)(i + j * TPF) * Cn + col;
        if (n < a.L) {
            const C conv = {v[j].x * invp, -v[j].y * invp};
            const C X = cmul(conv, chirp[n]);
            if (DST == 0) ((C*)a.spec_out)[(size_t)pair * a.L + n] = X;
            else a.xcorr[(size_t)pair * a.L + n] = X.x / (double)a.L;
        }
    }
}

__global__ void __launch_bounds__(256) any_chirp_kernel(double* __restrict__ chirp, int L) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= L) return;
    const unsigned long long r = ((unsigned long long)n * (unsigned long long)n) % (2ull * (unsigned long long)L);
    double s, c;
    sincospi(-(double)r / (double)L, &s, &c);                 // exp(-i pi n^2 / L), argument reduced exactly
    chirp[2 * n] = c;
    chirp[2 * n + 1] = s;
}

// the convolution's filter conj(c)[n] for |n| < L, wrapped to length P
__global__ void __launch_bounds__(256) any_filter_kernel(const double* __restrict__ chirp, double* __restrict__ b, int L, long long P) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= P) return;
    double re = 0.0, im = 0.0;
    if (n < L) { re = chirp[2 * n]; im = -chirp[2 * n + 1]; }
    else if (P - n < L) { re = chirp[2 * (P - n)]; im = -chirp[2 * (P - n) + 1]; }
    b[2 * n] = re;
    b[2 * n + 1] = im;
}

__global__ void __launch_bounds__(kGccThreads) any_means_kernel(const double* __restrict__ d0, const double* __restrict__ d1, int L,
                                                                double* __restrict__ means, unsigned long long* __restrict__ gmax) {
    __shared__ double red[16];
    const int pair = blockIdx.x, tid = threadIdx.x;
    for (int s = 0; s < 2; ++s) {
        const double* x = (s ? d1 : d0) + (size_t)pair * L;
        double acc = 0.0;
        for (int t = tid; t < L; t += kGccThreads) acc += x[t];
        const double m = block_sum(acc, red) / (double)L;
        if (tid == 0) means[2 * pair + s] = m;
    }
    if (tid == 0) gmax[pair] = 0ull;
}

// Z = D0 + i D1 in natural order -> G = conj(D0) D1 for k = 0..L/2 (kept in place of Z[k]) and its largest magnitude
__global__ void __launch_bounds__(256) any_cross_kernel(double* __restrict__ spec, int L, unsigned long long* __restrict__ gmax) {
    using C = cpx<double>;
    __shared__ double red[4];
    const int pair = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    C* Z = (C*)spec + (size_t)pair * L;
    double mag = 0.0;
    C g = {0.0, 0.0};
    C zk = {0.0, 0.0}, zm = {0.0, 0.0};
    const bool in = k <= L / 2;
    if (in) {
        zk = Z[k];
        zm = cconj(Z[k == 0 ? 0 : L - k]);
    }
    __syncthreads();
    if (in) {
        const C D0 = {0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y)};
        const C Dd = {0.5 * (zk.x - zm.x), 0.5 * (zk.y - zm.y)};      // i D1
        const C D1 = {Dd.y, -Dd.x};
        g = cmul(cconj(D0), D1);
        mag = hypot(g.x, g.y);
    }
    // every Z[k], Z[L-k] pair is read by exactly this thread (k <= L/2): safe to overwrite Z[k] now
    if (in) Z[k] = g;
    for (int o = 32; o > 0; o >>= 1) mag = fmax(mag, __shfl_down(mag, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mag;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(gmax + pair, (unsigned long long)__double_as_longlong(m));      // non-negative doubles order like their bits
    }
}

// PHAT weight, Hermitian extension, conjugated for the inverse: spec[k] = conj(Y[k]), spec[L-k] = Y[k]
__global__ void __launch_bounds__(256) any_weight_kernel(double* __restrict__ spec, int L, const unsigned long long* __restrict__ gmax) {
    using C = cpx<double>;
    const int pair = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > L / 2) return;
    C* Z = (C*)spec + (size_t)pair * L;
    const double m = __longlong_as_double((long long)gmax[pair]);
    C g = Z[k];
    const double w = 1.0 / (1e-10 * m + hypot(g.x, g.y));
    g = {g.x * w, g.y * w};
    if (k == 0 || k == L / 2) g.y = 0.0;                      // irfft ignores the imaginary part of the edge bins
    Z[k] = cconj(g);
    if (k != 0 && k != L / 2) Z[L - k] = g;
}

__global__ void __launch_bounds__(kGccThreads) any_argmax_kernel(const double* __restrict__ x, int L, int* __restrict__ argmax) {
    __shared__ double red[16];
    __shared__ int redi[16];
    const int tid = threadIdx.x, pair = blockIdx.x;
    const double* xp = x + (size_t)pair * L;
    double best = -1.0;
    int besti = 0;
    for (int t = tid; t < L; t += kGccThreads) {
        const double v = fabs(xp[t]);
        if (v > best) { best = v; besti = t; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_down(best, o, 64);
        const int oi = __shfl_down(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kGccThreads / 64; ++w)
            if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
        argmax[pair] = besti;
    }
}

template <int SRC>
static int launch_col_fwd(const AnyArgs& a, int pairs, hipStream_t s) {
    const dim3 grid((unsigned)(1u << a.log2c), pairs);
    switch (a.log2r) {
        case 8: hipLaunchKernelGGL((any_col_fwd_kernel<8, SRC>), grid, dim3(Pow2Plan<8>::TPF), 0, s, a); break;
        case 9: hipLaunchKernelGGL((any_col_fwd_kernel<9, SRC>), grid, dim3(Pow2Plan<9>::TPF), 0, s, a); break;
        case 10: hipLaunchKernelGGL((any_col_fwd_kernel<10, SRC>), grid, dim3(Pow2Plan<10>::TPF), 0, s, a); break;
        case 11: hipLaunchKernelGGL((any_col_fwd_kernel<11, SRC>), grid, dim3(Pow2Plan<11>::TPF), 0, s, a); break;
        case 12: hipLaunchKernelGGL((any_col_fwd_kernel<12, SRC>), grid, dim3(Pow2Plan<12>::TPF), 0, s, a); break;
        case 13: hipLaunchKernelGGL((any_col_fwd_kernel<13, SRC>), grid, dim3(Pow2Plan<13>::TPF), 0, s, a); break;
        default: set_last_error("gcc any-length: bad column size 2^%d", a.log2r); return FRT_ERR_UNSUPPORTED;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}
template <int MODE>
static int launch_row(const AnyArgs& a, int pairs, hipStream_t s) {
    const dim3 grid((unsigned)(1u << a.log2r), pairs);
    switch (a.log2c) {
        case 8: hipLaunchKernelGGL((any_row_kernel<8, MODE>), grid, dim3(Pow2Plan<8>::TPF), 0, s, a); break;
        case 9: hipLaunchKernelGGL((any_row_kernel<9, MODE>), grid, dim3(Pow2Plan<9>::TPF), 0, s, a); break;
        case 10: hipLaunchKernelGGL((any_row_kernel<10, MODE>), grid, dim3(Pow2Plan<10>::TPF), 0, s, a); break;
        case 11: hipLaunchKernelGGL((any_row_kernel<11, MODE>), grid, dim3(Pow2Plan<11>::TPF), 0, s, a); break;
        case 12: hipLaunchKernelGGL((any_row_kernel<12, MODE>), grid, dim3(Pow2Plan<12>::TPF), 0, s, a); break;
        case 13: hipLaunchKernelGGL((any_row_kernel<13, MODE>), grid, dim3(Pow2Plan<13>::TPF), 0, s, a); break;
        default: set_last_error("gcc any-length: bad row size 2^%d", a.log2c); return FRT_ERR_UNSUPPORTED;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}
template <int DST>
static int launch_col_inv(const AnyArgs& a, int pairs, hipStream_t s) {
    const dim3 grid((unsigned)(1u << a.log2c), pairs);
    switch (a.log2r) {
        case 8: hipLaunchKernelGGL((any_col_inv_kernel<8, DST>), grid, dim3(Pow2Plan<8>::TPF), 0, s, a); break;
        case 9: hipLaunchKernelGGL((any_col_inv_kernel<9, DST>), grid, dim3(Pow2Plan<9>::TPF), 0, s, a); break;
        case 10: hipLaunchKernelGGL((any_col_inv_kernel<10, DST>), grid, dim3(Pow2Plan<10>::TPF), 0, s, a); break;
        case 11: hipLaunchKernelGGL((any_col_inv_kernel<11, DST>), grid, dim3(Pow2Plan<11>::TPF), 0, s, a); break;
        case 12: hipLaunchKernelGGL((any_col_inv_kernel<12, DST>), grid, dim3(Pow2Plan<12>::TPF), 0, s, a); break;
        case 13: hipLaunchKernelGGL((any_col_inv_kernel<13, DST>), grid, dim3(Pow2Plan<13>::TPF), 0, s, a); break;
        default: set_last_error("gcc any-length: bad column size 2^%d", a.log2r); return FRT_ERR_UNSUPPORTED;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

}  // namespace frt

using namespace frt;

struct frt_gcc {
    int L = 0, M = 0, M2 = 0, R = 1, n_pairs = 0;
    MixedPlan plan{};
    hipStream_t stream = nullptr;
    DeviceBuffer window, twm, tw2, tws, twl, dw, scratch;
    int static_plan = 0;                 // 1: M2 = 6000, 2: M2 = 3000 — the compile-time plans of fft_static.h (gcc_fft)
    DeviceBuffer psum;
    DeviceBuffer in0, in1, out, argmax, means, old, sm, stats;
    size_t lds_bytes = 0;
    // any-length path (chirp-z): lengths the one-workgroup kernel does not take
    bool any = false;
    int log2r = 0, log2c = 0;
    DeviceBuffer chirp, bhat, work, spec, twr, twc, gmax, part_val, part_idx, prof;
};

extern "C" void frt_gcc_destroy(frt_gcc* h) {
    if (!h) return;
    free_retired_allocations(true);      // blocks parked by growing buffers (common.h); synchronises the device like the releases below
    DeviceBuffer* bufs[] = {&h->window, &h->twm, &h->tw2, &h->tws, &h->twl, &h->dw, &h->scratch, &h->in0, &h->in1,
                            &h->out, &h->argmax, &h->means, &h->old, &h->sm, &h->stats,
                            &h->chirp, &h->bhat, &h->work, &h->spec, &h->twr, &h->twc, &h->gmax, &h->part_val, &h->part_idx, &h->prof, &h->psum};
    for (auto* b : bufs) b->release();
    delete h;
}

extern "C" int frt_gcc_create(frt_gcc** out, int length, int n_pairs) {
    FRT_REQUIRE(out, "frt_gcc_create: null handle pointer");
    *out = nullptr;
    FRT_REQUIRE(length >= 4 && length % 2 == 0, "frt_gcc_create: length %d must be even and >= 4", length);
    FRT_REQUIRE(n_pairs >= 1, "frt_gcc_create: n_pairs %d < 1", n_pairs);
    frt_gcc* h = new frt_gcc();
    h->L = length;
    h->M = length / 2;
    h->n_pairs = n_pairs;
    int R = 1;
    while (h->M / R > kGccMaxM2 && R < kGccMaxR && h->M % (2 * R) == 0) R *= 2;
    if (const char* fr = exp_env("FRT_GCC_FORCE_R")) {           // experiments: a deeper split than LDS requires
        const int want = atoi(fr);
        while (R < want && R < kGccMaxR && h->M % (2 * R) == 0) R *= 2;
    }
    // The default window (24000 samples: M = 12000) in batches whose eight forward workgroups per pair find a CU each (32 pairs on 256
    // CUs; the widget's one pair): four sub-transforms of 3000 points instead of two of 6000 — eight forward and four inverse workgroups
    // per pair side by side.  Measured (profiles/r05_gcc_batch.txt): 1 pair 47.5 -> 35.1 us, 4: 50.1 -> 36.7, 16: 58.1 -> 45.8,
    // 32: 62.6 -> 56.7; beyond that the quarter-strided loads and stores of the four-way split cost more than its width buys
    // (64 pairs 72 -> 80 us, 100: 84 -> 114) and the two-way split stays.
    const bool small_batch = h->M == 2 * kGccStaticM2 && (long long)n_pairs * 8 <= (long long)device_cu_count() && option(kOptGccOneWorkgroup) <= 0 &&
                             exp_env("FRT_GCC_NO_STATIC_PLAN") == nullptr && exp_env("FRT_GCC_NO_SMALL_PLAN") == nullptr;
    if (small_batch && R == 2) R = 4;
    h->R = R;
    h->M2 = h->M / R;
    // numpy.hanning(L) = 0.5 - 0.5 cos(2 pi n / (L - 1))
    std::vector<double> win(length);
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < length; ++n) win[n] = 0.5 - 0.5 * std::cos(2.0 * pi * n / (length - 1));
    int rc;
    if (h->M2 > kGccMaxM2 || !make_mixed_plan(h->M2, &h->plan) || option(kOptGccAnyLength) > 0) {
        // any other length (numpy's rfft takes them all): chirp-z on a four-step power-of-two transform
        FRT_REQUIRE(length <= (1 << 25), "frt_gcc_create: length %d above 2^25 samples", length);
        h->any = true;
        int p = 16;
        while ((1ll << p) < 2ll * length - 1) ++p;
        h->log2r = p / 2;
        h->log2c = p - h->log2r;
        const long long P = 1ll << p;
        if ((rc = upload(h->window, win)) || (rc = upload(h->twr, make_twiddles<double>(1 << h->log2r))) ||
            (rc = upload(h->twc, make_twiddles<double>(1 << h->log2c))) || (rc = h->chirp.reserve((size_t)length * 16)) ||
            (rc = h->bhat.reserve((size_t)P * 16)) || (rc = h->work.reserve((size_t)n_pairs * P * 16)) ||
            (rc = h->spec.reserve((size_t)n_pairs * length * 16)) || (rc = h->gmax.reserve((size_t)n_pairs * 8))) {
            frt_gcc_destroy(h);
            return rc;
        }
        // chirp, and the transform of the convolution's filter in the transposed layout of the four-step scheme
        hipLaunchKernelGGL(any_chirp_kernel, dim3((length + 255) / 256), dim3(256), 0, nullptr, h->chirp.as<double>(), length);
        hipLaunchKernelGGL(any_filter_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, nullptr, h->chirp.as<double>(),
                           h->work.as<double>(), length, P);
        AnyArgs a{};
        a.L = length;
        a.log2r = h->log2r;
        a.log2c = h->log2c;
        a.P = P;
        a.plain = h->work.as<double>();
        a.work = h->bhat.as<double>();
        a.twr = h->twr.as<double>();
        a.twc = h->twc.as<double>();
        a.chirp = h->chirp.as<double>();
        if ((rc = launch_col_fwd<0>(a, 1, nullptr)) || (rc = launch_row<0>(a, 1, nullptr))) {
            frt_gcc_destroy(h);
            return rc;
        }
        if (hipDeviceSynchronize() != hipSuccess) {
            set_last_error("frt_gcc_create: chirp-z tables failed");
            frt_gcc_destroy(h);
            return FRT_ERR_HIP;
        }
        *out = h;
        return FRT_OK;
    }
    if ((rc = upload(h->window, win)) || (rc = upload(h->twm, make_twiddles<double>(h->M))) ||
        (rc = upload(h->tw2, make_pass_twiddles<double>(h->plan))) || (rc = upload(h->twl, make_twiddles<double>(length, h->M + 1)))) {
        frt_gcc_destroy(h);
        return rc;
    }
    // rfft(window): the means' contribution to the spectra (gcc_phat_kernel subtracts mean * dw instead of the mean itself)
    if ((rc = h->dw.reserve((size_t)(h->M + 1) * 16))) {
        frt_gcc_destroy(h);
        return rc;
    }
    hipLaunchKernelGGL(gcc_window_rfft_kernel, dim3((h->M + 1 + 255) / 256), dim3(256), 0, nullptr, h->window.as<double>(), h->twl.as<double>(),
                       length, h->M, h->dw.as<double>());
    if (hipDeviceSynchronize() != hipSuccess) {
        set_last_error("frt_gcc_create: the window's spectrum failed");
        frt_gcc_destroy(h);
        return FRT_ERR_HIP;
    }
    h->lds_bytes = (size_t)h->M2 * 16 + 16 * sizeof(double) + 16 * sizeof(int);
    h->static_plan = exp_env("FRT_GCC_NO_STATIC_PLAN") != nullptr ? 0 : h->M2 == kGccStaticM2 ? 1 : (h->M2 == kGccStaticM2Small && small_batch) ? 2 : 0;
    if (h->static_plan == 1) rc = upload(h->tws, make_static_twiddles<double>({6, 10, 10, 10}));
    else if (h->static_plan == 2 && !(rc = upload(h->tws, make_static_twiddles<double>({3, 10, 10, 10}))))
        rc = h->psum.reserve((size_t)n_pairs * 2 * 4 * sizeof(double));
    if (rc) {
        frt_gcc_destroy(h);
        return rc;
    }
    const void* big[] = {(const void*)gcc_phat_kernel<1, 0>, (const void*)gcc_phat_kernel<2, 0>, (const void*)gcc_phat_kernel<4, 0>,
                         (const void*)gcc_phat_kernel<2, 1>, (const void*)gcc_fwd_kernel<1, 0>,  (const void*)gcc_fwd_kernel<2, 0>,
                         (const void*)gcc_fwd_kernel<4, 0>,  (const void*)gcc_fwd_kernel<2, 1>,  (const void*)gcc_fwd_kernel<4, 2>,
                         (const void*)gcc_inv_kernel<1, 0>,  (const void*)gcc_inv_kernel<2, 0>,  (const void*)gcc_inv_kernel<4, 0>,
                         (const void*)gcc_inv_kernel<2, 1>,  (const void*)gcc_inv_kernel<4, 2>};
    for (const void* f : big)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) {
            set_last_error("frt_gcc_create: cannot reserve %zu bytes of LDS", h->lds_bytes);
            frt_gcc_destroy(h);
            return FRT_ERR_HIP;
        }
    if (h->static_plan == 1 && hipFuncSetAttribute((const void*)gcc_phat_resident_kernel<kResThreads>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)ResPlan<kResThreads>::LDS_BYTES) != hipSuccess) {
        set_last_error("frt_gcc_create: cannot reserve %zu bytes of LDS", ResPlan<kResThreads>::LDS_BYTES);
        frt_gcc_destroy(h);
        return FRT_ERR_HIP;
    }
    *out = h;
    return FRT_OK;
}

extern "C" int frt_gcc_set_stream(frt_gcc* h, void* s) {
    FRT_REQUIRE(h, "frt_gcc_set_stream: null handle");
    h->stream = (hipStream_t)s;
    return FRT_OK;
}

extern "C" int frt_gcc_phat(frt_gcc* h, const double* d0, const double* d1, double* xcorr_out, int* argmax_out, double* means_out) {
    FRT_REQUIRE(h && d0 && d1 && xcorr_out, "frt_gcc_phat: null argument");
    const bool dev = is_device_pointer(d0);
    FRT_REQUIRE(dev == is_device_pointer(d1) && dev == is_device_pointer(xcorr_out),
                "frt_gcc_phat: buffers must all be host or all be device memory");
    const size_t bytes = (size_t)h->n_pairs * h->L * sizeof(double);
    GccArgs a{};
    int rc;
    if ((rc = h->argmax.reserve(h->n_pairs * sizeof(int))) || (rc = h->means.reserve(h->n_pairs * 2 * sizeof(double)))) return rc;
    if (dev) {
        a.d0 = d0;
        a.d1 = d1;
        a.xcorr = xcorr_out;
    } else {
        if ((rc = h->in0.reserve(bytes)) || (rc = h->in1.reserve(bytes)) || (rc = h->out.reserve(bytes))) return rc;
        FRT_HIP_CHECK(hipMemcpyAsync(h->in0.ptr, d0, bytes, hipMemcpyHostToDevice, h->stream));
        FRT_HIP_CHECK(hipMemcpyAsync(h->in1.ptr, d1, bytes, hipMemcpyHostToDevice, h->stream));
        a.d0 = h->in0.as<double>();
        a.d1 = h->in1.as<double>();
        a.xcorr = h->out.as<double>();
    }
    if (h->any) {
        AnyArgs q{};
        q.L = h->L;
        q.log2r = h->log2r;
        q.log2c = h->log2c;
        q.P = 1ll << (h->log2r + h->log2c);
        q.d0 = a.d0;
        q.d1 = a.d1;
        q.means = h->means.as<double>();
        q.window = h->window.as<double>();
        q.chirp = h->chirp.as<double>();
        q.spec = h->spec.as<double>();
        q.work = h->work.as<double>();
        q.bhat = h->bhat.as<double>();
        q.twr = h->twr.as<double>();
        q.twc = h->twc.as<double>();
        q.spec_out = h->spec.as<double>();
        q.xcorr = a.xcorr;
        unsigned long long* gm = h->gmax.as<unsigned long long>();
        const dim3 halfgrid((h->L / 2 + 1 + 255) / 256, h->n_pairs);
        hipLaunchKernelGGL(any_means_kernel, dim3(h->n_pairs), dim3(kGccThreads), 0, h->stream, q.d0, q.d1, h->L, h->means.as<double>(), gm);
        if ((rc = launch_col_fwd<1>(q, h->n_pairs, h->stream)) || (rc = launch_row<1>(q, h->n_pairs, h->stream)) ||
            (rc = launch_col_inv<0>(q, h->n_pairs, h->stream)))
            return rc;
        hipLaunchKernelGGL(any_cross_kernel, halfgrid, dim3(256), 0, h->stream, h->spec.as<double>(), h->L, gm);
        hipLaunchKernelGGL(any_weight_kernel, halfgrid, dim3(256), 0, h->stream, h->spec.as<double>(), h->L, gm);
        if ((rc = launch_col_fwd<2>(q, h->n_pairs, h->stream)) || (rc = launch_row<1>(q, h->n_pairs, h->stream)) ||
            (rc = launch_col_inv<1>(q, h->n_pairs, h->stream)))
            return rc;
        hipLaunchKernelGGL(any_argmax_kernel, dim3(h->n_pairs), dim3(kGccThreads), 0, h->stream, a.xcorr, h->L, h->argmax.as<int>());
        FRT_HIP_CHECK(hipGetLastError());
    } else {
    a.argmax = h->argmax.as<int>();
    a.means = h->means.as<double>();
    a.window = h->window.as<double>();
    a.twm = h->twm.as<double>();
    a.tw2 = h->tw2.as<double>();
    a.twl = h->twl.as<double>();
    a.plan = h->plan;
    a.L = h->L;
    a.M = h->M;
    a.M2 = h->M2;
    a.R = h->R;
    a.tws = h->tws.as<double>();
    a.dw = h->dw.as<double>();
    static const bool profile = exp_env("FRT_GCC_PROFILE") != nullptr;
    if (profile) {
        if ((rc = h->prof.reserve(16 * sizeof(long long)))) return rc;
        a.prof = h->prof.as<long long>();
    }
    a.vec = ((uintptr_t)a.d0 % 16 == 0) && ((uintptr_t)a.d1 % 16 == 0) && ((uintptr_t)a.xcorr % 16 == 0);
    const int st = h->static_plan;
    a.psum = st == 2 ? h->psum.as<double>() : nullptr;
    const int force = option(kOptGccOneWorkgroup);
    // the default window, one workgroup per pair: nothing passes through HBM between the signals and the correlation (gcc_resident.h)
    const bool can_reside = st == 1 && h->R == 2 && option(kOptGccResident) != 0;
    // A pair as launches of its own phases while the batch leaves most CUs idle: below a sixth of a workgroup per CU when the
    // resident kernel serves the window (measured, profiles/r06_gcc_batch.txt, us per call: 32 pairs 57 against 64, 40 pairs 65 = 65,
    // 48 pairs 67 against 65, 64 pairs 72 against 67, 100 pairs 84 against 71, 160 pairs 135 against 75), below 5/8 with the slab kernel
    // (round 3: crossover between 100 and 256 pairs)
    const bool split = force >= 0 ? force == 0 : can_reside ? (long long)h->n_pairs * 6 < (long long)device_cu_count()
                                                             : (long long)h->n_pairs * 8 <= 5ll * device_cu_count();
    const bool resident = !split && can_reside;
    if (!resident && (rc = h->scratch.reserve((size_t)h->n_pairs * (4 * (size_t)h->M + 2) * 2 * sizeof(double)))) return rc;
    a.scratch = h->scratch.as<double>();
    if (split) {
        if ((rc = h->gmax.reserve((size_t)h->n_pairs * 8))) return rc;
        unsigned long long* gm = h->gmax.as<unsigned long long>();
        if ((rc = h->part_val.reserve((size_t)h->n_pairs * h->R * sizeof(double))) || (rc = h->part_idx.reserve((size_t)h->n_pairs * h->R * sizeof(int))))
            return rc;
        const dim3 fgrid(h->R <= 2 ? 2 : 2 * h->R, h->n_pairs), igrid(h->R, h->n_pairs), cgrid((h->M / 2 + 1 + 255) / 256, h->n_pairs);
        const dim3 block(kGccThreads);
        if (st == 1) hipLaunchKernelGGL((gcc_fwd_kernel<2, 1>), fgrid, block, h->lds_bytes, h->stream, a, gm);
        else if (st == 2) hipLaunchKernelGGL((gcc_fwd_kernel<4, 2>), fgrid, block, h->lds_bytes, h->stream, a, gm);
        else if (h->R == 1) hipLaunchKernelGGL((gcc_fwd_kernel<1, 0>), fgrid, block, h->lds_bytes, h->stream, a, gm);
        else if (h->R == 2) hipLaunchKernelGGL((gcc_fwd_kernel<2, 0>), fgrid, block, h->lds_bytes, h->stream, a, gm);
        else hipLaunchKernelGGL((gcc_fwd_kernel<4, 0>), fgrid, block, h->lds_bytes, h->stream, a, gm);
        if (h->R == 1) hipLaunchKernelGGL((gcc_cross_kernel<1>), cgrid, dim3(256), 0, h->stream, a, gm);
        else if (h->R == 2) hipLaunchKernelGGL((gcc_cross_kernel<2>), cgrid, dim3(256), 0, h->stream, a, gm);
        else if (st == 2) hipLaunchKernelGGL((gcc_cross_kernel<4, true>), cgrid, dim3(256), 0, h->stream, a, gm);
        else hipLaunchKernelGGL((gcc_cross_kernel<4>), cgrid, dim3(256), 0, h->stream, a, gm);
        hipLaunchKernelGGL(gcc_pack_kernel, cgrid, dim3(256), 0, h->stream, a, gm);
        double* pv = h->part_val.as<double>();
        int* pi = h->part_idx.as<int>();
        if (st == 1) hipLaunchKernelGGL((gcc_inv_kernel<2, 1>), igrid, block, h->lds_bytes, h->stream, a, pv, pi);
        else if (st == 2) hipLaunchKernelGGL((gcc_inv_kernel<4, 2>), igrid, block, h->lds_bytes, h->stream, a, pv, pi);
        else if (h->R == 1) hipLaunchKernelGGL((gcc_inv_kernel<1, 0>), igrid, block, h->lds_bytes, h->stream, a, pv, pi);
        else if (h->R == 2) hipLaunchKernelGGL((gcc_inv_kernel<2, 0>), igrid, block, h->lds_bytes, h->stream, a, pv, pi);
        else hipLaunchKernelGGL((gcc_inv_kernel<4, 0>), igrid, block, h->lds_bytes, h->stream, a, pv, pi);
        hipLaunchKernelGGL(gcc_argmax_combine_kernel, dim3((h->n_pairs + 255) / 256), dim3(256), 0, h->stream, h->part_val.as<double>(),
                           h->part_idx.as<int>(), h->R, h->n_pairs, h->argmax.as<int>());
    } else {
        const dim3 grid(h->n_pairs), block(kGccThreads);
        if (resident) {
            hipLaunchKernelGGL(gcc_phat_resident_kernel<kResThreads>, grid, dim3(kResThreads), ResPlan<kResThreads>::LDS_BYTES, h->stream, a);
        } else if (st == 1) hipLaunchKernelGGL((gcc_phat_kernel<2, 1>), grid, block, h->lds_bytes, h->stream, a);
        else if (h->R == 1) hipLaunchKernelGGL((gcc_phat_kernel<1, 0>), grid, block, h->lds_bytes, h->stream, a);
        else if (h->R == 2) hipLaunchKernelGGL((gcc_phat_kernel<2, 0>), grid, block, h->lds_bytes, h->stream, a);
        else hipLaunchKernelGGL((gcc_phat_kernel<4, 0>), grid, block, h->lds_bytes, h->stream, a);
    }
    FRT_HIP_CHECK(hipGetLastError());
    if (profile && !split) {
        long long t[9];
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        FRT_HIP_CHECK(hipMemcpy(t, h->prof.ptr, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, resident ? "gcc_phat_resident_kernel phases (us): load of signal 0 %.1f | its first transform %.1f | quads, second transform %.1f | signal 1 %.1f | cross %.1f | pack %.1f | inverse transforms %.1f | stores, argmax %.1f | total %.1f\n"
                                 : "gcc_phat_kernel phases (us): means %.1f | load0 %.1f | fft0 %.1f | rest of forward %.1f | cross %.1f | pack %.1f | inverse %.1f | argmax %.1f | total %.1f\n",
                (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01, (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01,
                (t[7] - t[6]) * 0.01, (t[8] - t[7]) * 0.01, (t[8] - t[0]) * 0.01);
    }
    }
    if (!dev) FRT_HIP_CHECK(hipMemcpyAsync(xcorr_out, h->out.ptr, bytes, hipMemcpyDeviceToHost, h->stream));
    if (argmax_out) {
        const bool adev = is_device_pointer(argmax_out);
        FRT_HIP_CHECK(hipMemcpyAsync(argmax_out, h->argmax.ptr, h->n_pairs * sizeof(int),
                                     adev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    }
    if (means_out) {
        const bool mdev = is_device_pointer(means_out);
        FRT_HIP_CHECK(hipMemcpyAsync(means_out, h->means.ptr, h->n_pairs * 2 * sizeof(double),
                                     mdev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    }
    if (!dev || (argmax_out && !is_device_pointer(argmax_out)) || (means_out && !is_device_pointer(means_out)))
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    return FRT_OK;
}

extern "C" int frt_gcc_readout(frt_gcc* h, const double* xcorr, const double* old_smoothed, double alpha, double sample_rate,
                               double delayrange_s, double* smoothed_out, frt_delay_readout* readout) {
    FRT_REQUIRE(h && xcorr && smoothed_out && readout, "frt_gcc_readout: null argument");
    FRT_REQUIRE(sample_rate > 0, "frt_gcc_readout: sample_rate must be positive");
    const bool dev = is_device_pointer(xcorr);
    FRT_REQUIRE(dev == is_device_pointer(smoothed_out) && (!old_smoothed || dev == is_device_pointer(old_smoothed)),
                "frt_gcc_readout: buffers must all be host or all be device memory");
    const size_t bytes = (size_t)h->n_pairs * h->L * sizeof(double);
    int rc;
    if ((rc = h->argmax.reserve(h->n_pairs * sizeof(int))) || (rc = h->stats.reserve(h->n_pairs * 2 * sizeof(double)))) return rc;
    const double* dx = xcorr;
    const double* dold = old_smoothed;
    double* dsm = smoothed_out;
    if (!dev) {
        if ((rc = h->out.reserve(bytes)) || (rc = h->sm.reserve(bytes))) return rc;
        FRT_HIP_CHECK(hipMemcpyAsync(h->out.ptr, xcorr, bytes, hipMemcpyHostToDevice, h->stream));
        dx = h->out.as<double>();
        dsm = h->sm.as<double>();
        if (old_smoothed) {
            if ((rc = h->old.reserve(bytes))) return rc;
            FRT_HIP_CHECK(hipMemcpyAsync(h->old.ptr, old_smoothed, bytes, hipMemcpyHostToDevice, h->stream));
            dold = h->old.as<double>();
        }
    }
    hipLaunchKernelGGL(gcc_readout_kernel, dim3(h->n_pairs), dim3(kGccThreads), 0, h->stream, dx, dold, dsm, h->L, alpha,
                       h->argmax.as<int>(), h->stats.as<double>());
    FRT_HIP_CHECK(hipGetLastError());
    if (!dev) FRT_HIP_CHECK(hipMemcpyAsync(smoothed_out, dsm, bytes, hipMemcpyDeviceToHost, h->stream));
    std::vector<int> am(h->n_pairs);
    std::vector<double> st(2 * h->n_pairs);
    FRT_HIP_CHECK(hipMemcpyAsync(am.data(), h->argmax.ptr, am.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    FRT_HIP_CHECK(hipMemcpyAsync(st.data(), h->stats.ptr, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int p = 0; p < h->n_pairs; ++p) {           // scalar tail of delay_estimator.py:141-176
        frt_delay_readout& r = readout[p];
        r.argmax = am[p];
        r.extremum = st[2 * p];
        const double peak_norm = std::fabs(st[2 * p]) / (3.0 * st[2 * p + 1]);
        const double time = 2.0 * delayrange_s;
        double delay_ms = 1e3 * (double)am[p] / sample_rate;
        if (delay_ms > 1e3 * time / 2.0) delay_ms -= 1e3 * time;
        r.delay_ms = delay_ms;
        r.distance_m = delay_ms * 1e-3 * 340.0;
        double x = peak_norm > 1.0 ? peak_norm - 1.0 : 0.0;
        x = std::pow(0.12 * x, 3.0);
        r.correlation_pct = (int)((x / (1.0 + x)) * 100.0);
    }
    return FRT_OK;
}
