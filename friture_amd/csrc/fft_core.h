// fft_core.h — device-side FFT building blocks for gfx950 (CDNA4, wave64).
//
// Everything in the Friture hot path is an FFT of a real signal (audioproc.py:44,
// filter.py:206-220, correlation.py:34-41 in the reference).  A real N-point transform is
// computed here as an M = N/2 point complex transform of z[n] = x[2n] + i x[2n+1] followed by
// a conjugate-symmetric "unpack" step.
//
// Layout of one power-of-two complex transform:
//   * every thread owns E = 8 complex points; a frame is served by TPF = M/8 threads;
//   * passes are Stockham autosort passes (natural order in, natural order out) of radix
//     8, 8, ..., {1|2|4}; a pass with radix R < 8 performs 8/R butterflies per thread;
//   * in every pass thread i reads the points  i + j*TPF  (j = 0..7): consecutive lanes touch
//     consecutive LDS addresses (conflict free), and the first pass takes them straight from the
//     registers the HBM loads landed in;
//   * the scatter side of each pass goes through LDS with one pad slot every 8 points, which
//     makes the stride-8 scatter of the first pass conflict free for ds_write_b64.
#pragma once
#include <hip/hip_runtime.h>

namespace frt {

template <typename T>
struct alignas(2 * sizeof(T)) cpx {
    T x, y;
};

template <typename T>
__device__ __forceinline__ cpx<T> operator+(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cpx<T> operator-(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
__device__ __forceinline__ cpx<T> cconj(cpx<T> a) { return {a.x, -a.y}; }
// multiply by -i
template <typename T>
__device__ __forceinline__ cpx<T> mul_mi(cpx<T> a) { return {a.y, -a.x}; }
// multiply by +i
template <typename T>
__device__ __forceinline__ cpx<T> mul_pi(cpx<T> a) { return {-a.y, a.x}; }

// ---- forward DFT butterflies, natural-order output -------------------------------------------
template <typename T>
__device__ __forceinline__ void dft2(cpx<T>& a, cpx<T>& b) {
    cpx<T> t = a;
    a = t + b;
    b = t - b;
}

template <typename T>
__device__ __forceinline__ void dft4(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2, cpx<T>& a3) {
    cpx<T> s0 = a0 + a2, s1 = a0 - a2, s2 = a1 + a3, s3 = mul_mi(a1 - a3);
    a0 = s0 + s2;
    a1 = s1 + s3;
    a2 = s0 - s2;
    a3 = s1 - s3;
}

template <typename T>
__device__ __forceinline__ void dft8(cpx<T> (&a)[8]) {
    const T h = (T)0.70710678118654752440;
    // split n = m + 4 s: even outputs are DFT4 of sums, odd outputs DFT4 of twiddled differences
    cpx<T> e0 = a[0] + a[4], e1 = a[1] + a[5], e2 = a[2] + a[6], e3 = a[3] + a[7];
    cpx<T> d0 = a[0] - a[4], d1 = a[1] - a[5], d2 = a[2] - a[6], d3 = a[3] - a[7];
    // W8^1 = h(1 - i), W8^2 = -i, W8^3 = -h(1 + i)
    cpx<T> o1 = {h * (d1.x + d1.y), h * (d1.y - d1.x)};
    cpx<T> o2 = mul_mi(d2);
    cpx<T> o3 = {h * (d3.y - d3.x), -h * (d3.x + d3.y)};
    dft4(e0, e1, e2, e3);
    dft4(d0, o1, o2, o3);
    a[0] = e0; a[2] = e1; a[4] = e2; a[6] = e3;
    a[1] = d0; a[3] = o1; a[5] = o2; a[7] = o3;
}

// ---- inverse DFT butterflies (conjugate twiddles), natural-order output ----------------------
template <typename T>
__device__ __forceinline__ void idft4(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2, cpx<T>& a3) {
    cpx<T> s0 = a0 + a2, s1 = a0 - a2, s2 = a1 + a3, s3 = mul_pi(a1 - a3);
    a0 = s0 + s2;
    a1 = s1 + s3;
    a2 = s0 - s2;
    a3 = s1 - s3;
}

// One pad slot per 8 points (see header comment).
__device__ __forceinline__ int lds_pad(int idx) { return idx + (idx >> 3); }
constexpr int lds_padded_size(int m) { return m + (m >> 3); }

// Compile-time description of the pass schedule of an M-point transform (M = 2^LOG2M >= 8).
template <int LOG2M>
struct Pow2Plan {
    static constexpr int M = 1 << LOG2M;
    static constexpr int TPF = M / 8;                       // threads per frame
    static constexpr int NP8 = LOG2M / 3;                    // radix-8 passes
    static constexpr int RLAST = 1 << (LOG2M - 3 * NP8);     // 1, 2 or 4
    static constexpr int NPASS = NP8 + (RLAST > 1 ? 1 : 0);
    // number of per-thread twiddle factors over all passes after the first
    static constexpr int NTW = 7 * (NP8 - 1) + (RLAST == 2 ? 4 : RLAST == 4 ? 6 : 0);
};

// Twiddle providers.  The factors of every pass after the first depend only on the thread's index
// inside its frame.  TwRegs fetches them once (callers hoist it out of frame loops: right for the
// one-wave-per-frame kernels, which have registers to spare); TwTable re-reads them from the table
// tw[n] = exp(-2 pi i n / M) at every use (right for the many-wave-per-frame kernels, whose
// workgroup size caps the register budget).  `zero` is an opaque 0 that keeps the compiler from
// hoisting those loads back out of the frame loop.
template <typename T, int LOG2M>
struct TwTable {
    const cpx<T>* tw;
    int zero;
    __device__ __forceinline__ cpx<T> get(int /*n*/, int idx) const { return tw[idx + zero]; }
};

template <typename T, int LOG2M>
struct TwRegs {
    static constexpr int NTW = Pow2Plan<LOG2M>::NTW > 0 ? Pow2Plan<LOG2M>::NTW : 1;
    cpx<T> w[NTW];
    __device__ __forceinline__ cpx<T> get(int n, int /*idx*/) const { return w[n]; }
    // i = thread index inside the frame
    __device__ __forceinline__ void load(const cpx<T>* __restrict__ tw, int i) {
        using P = Pow2Plan<LOG2M>;
        int n = 0;
        int p = 8;
#pragma unroll
        for (int pass = 1; pass < P::NP8; ++pass) {
            const int k = i & (p - 1);
            const int step = P::M / (8 * p);      // W_{8p}^{qk} = W_M^{qk * M/(8p)}
#pragma unroll
            for (int q = 1; q < 8; ++q) w[n++] = tw[(q * k * step) & (P::M - 1)];
            p *= 8;
        }
        if (P::RLAST == 4) {
            const int step = P::M / (4 * p);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int k = (i + c * P::TPF) & (p - 1);
#pragma unroll
                for (int q = 1; q < 4; ++q) w[n++] = tw[(q * k * step) & (P::M - 1)];
            }
        } else if (P::RLAST == 2) {
            const int step = P::M / (2 * p);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = (i + c * P::TPF) & (p - 1);
                w[n++] = tw[(k * step) & (P::M - 1)];
            }
        }
    }
};

// Synchronisation between the scatter of one pass and the gather of the next.  When a frame lives
// inside one wavefront (TPF <= 64) LDS traffic of a wave is already ordered by the hardware, so a
// compiler-level fence is all that is needed; otherwise the frame's waves meet at a barrier.
template <bool WAVE_LOCAL>
__device__ __forceinline__ void pass_sync() {
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Forward M-point complex FFT of the 8 points v[j] = z[i + j*TPF] held by thread i of a frame.
// On return v[j] = Z[i + j*TPF].  `buf` is the frame's LDS scratch of lds_padded_size(M) points.
template <typename T, int LOG2M, bool WAVE_LOCAL, typename TW>
__device__ __forceinline__ void fft_pow2_forward(cpx<T> (&v)[8], cpx<T>* buf, int i, const TW& tw) {
    using P = Pow2Plan<LOG2M>;
    constexpr int TPF = P::TPF, M = P::M;
    int n = 0;
    int p = 1;
#pragma unroll
    for (int pass = 0; pass < P::NP8; ++pass) {
        const int k = i & (p - 1);
        if (pass > 0) {
            const int step = M / (8 * p);
#pragma unroll
            for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw.get(n++, (q * k * step) & (M - 1)));
        }
        dft8(v);
        const bool last = (pass == P::NPASS - 1);
        if (!last) {
            const int base = (i - k) * 8 + k;
            if (pass > 0) pass_sync<WAVE_LOCAL>();      // everyone has gathered the previous pass
#pragma unroll
            for (int q = 0; q < 8; ++q) buf[lds_pad(base + q * p)] = v[q];
            pass_sync<WAVE_LOCAL>();
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = buf[lds_pad(i + j * TPF)];
        }
        p *= 8;
    }
    if constexpr (P::RLAST == 4) {
        // two radix-4 butterflies: butterfly c works on slots c + 2q; it is always the last pass
        const int step = M / (4 * p);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = (i + c * TPF) & (p - 1);
#pragma unroll
            for (int q = 1; q < 4; ++q)
                v[c + 2 * q] = cmul(v[c + 2 * q], tw.get(n++, (q * k * step) & (M - 1)));
            dft4(v[c], v[c + 2], v[c + 4], v[c + 6]);
        }
    } else if constexpr (P::RLAST == 2) {
        const int step = M / (2 * p);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = (i + c * TPF) & (p - 1);
            v[c + 4] = cmul(v[c + 4], tw.get(n++, (k * step) & (M - 1)));
            dft2(v[c], v[c + 4]);
        }
    }
}

}  // namespace frt
