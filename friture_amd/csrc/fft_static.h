// fft_static.h — workgroup-wide complex FFT of a COMPILE-TIME 5-smooth length in LDS, radices up to 10 (gfx950).
//
// fft_mixed.h serves any 5-smooth length with run-time radices 4 / 5 / 3 / 2: for the delay estimator's default window
// (24000 samples -> two 6000-point complex sub-transforms, friture/delay_estimator.py:114-115) that is six passes, each
// with a branch on the radix, up to four twiddle gathers per butterfly from global memory and two barriers — ~30 us per
// transform, most of GCC-PHAT's time.  Here the plan is a template: 6000 = 6 x 10 x 10 x 10 is FOUR Stockham passes whose
// index arithmetic the compiler folds to constants, and a butterfly fetches ONE twiddle (exp(-2 pi i k / (p R)) from the
// pass's table of p entries) and raises it to the powers 2 .. R-1 itself.  Same pass scheme as fft_mixed.h: in a pass of
// radix R with p = product of the earlier radices, butterfly j < n/R reads the points j + q n/R, multiplies point q by
// w^q (k = j mod p), takes the R-point DFT and writes to (j - k) R + k + q p; gather - barrier - scatter - barrier, in place.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "fft_core.h"
#include "fft_mixed.h"

namespace frt {

// X[k] = sum_n a[n] exp(-2 pi i n k / 6), in place
template <typename T>
__device__ __forceinline__ void dft6(cpx<T> (&a)[6]) {
    dft3(a[0], a[2], a[4]);
    dft3(a[1], a[3], a[5]);
    const T h = (T)0.5, s = (T)0.86602540378443864676;
    const cpx<T> o1 = {h * a[3].x + s * a[3].y, h * a[3].y - s * a[3].x};       // a[3] exp(-i pi / 3)
    const cpx<T> o2 = {-h * a[5].x + s * a[5].y, -h * a[5].y - s * a[5].x};     // a[5] exp(-2 i pi / 3)
    const cpx<T> e0 = a[0], e1 = a[2], e2 = a[4], o0 = a[1];
    a[0] = e0 + o0;
    a[3] = e0 - o0;
    a[1] = e1 + o1;
    a[4] = e1 - o1;
    a[2] = e2 + o2;
    a[5] = e2 - o2;
}

// X[k] = sum_n a[n] exp(-2 pi i n k / 10), in place
template <typename T>
__device__ __forceinline__ void dft10(cpx<T> (&a)[10]) {
    dft5(a[0], a[2], a[4], a[6], a[8]);
    dft5(a[1], a[3], a[5], a[7], a[9]);
    // odd half times exp(-2 pi i k / 10), k = 1 .. 4
    const T c1 = (T)0.80901699437494742410, s1 = (T)0.58778525229247312917;     // cos / sin (pi / 5)
    const T c2 = (T)0.30901699437494742410, s2 = (T)0.95105651629515357212;     // cos / sin (2 pi / 5)
    const cpx<T> o0 = a[1];
    const cpx<T> o1 = {c1 * a[3].x + s1 * a[3].y, c1 * a[3].y - s1 * a[3].x};
    const cpx<T> o2 = {c2 * a[5].x + s2 * a[5].y, c2 * a[5].y - s2 * a[5].x};
    const cpx<T> o3 = {-c2 * a[7].x + s2 * a[7].y, -c2 * a[7].y - s2 * a[7].x};
    const cpx<T> o4 = {-c1 * a[9].x + s1 * a[9].y, -c1 * a[9].y - s1 * a[9].x};
    const cpx<T> e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], e4 = a[8];
    a[0] = e0 + o0;
    a[5] = e0 - o0;
    a[1] = e1 + o1;
    a[6] = e1 - o1;
    a[2] = e2 + o2;
    a[7] = e2 - o2;
    a[3] = e3 + o3;
    a[8] = e3 - o3;
    a[4] = e4 + o4;
    a[9] = e4 - o4;
}

template <typename T, int R>
__device__ __forceinline__ void dft_static(cpx<T> (&a)[R]) {
    static_assert(R == 2 || R == 3 || R == 4 || R == 5 || R == 6 || R == 10, "radix");
    if constexpr (R == 2) dft2(a[0], a[1]);
    if constexpr (R == 3) dft3(a[0], a[1], a[2]);
    if constexpr (R == 4) dft4(a[0], a[1], a[2], a[3]);
    if constexpr (R == 5) dft5(a[0], a[1], a[2], a[3], a[4]);
    if constexpr (R == 6) dft6(a);
    if constexpr (R == 10) dft10(a);
}

// One pass.  tw: the pass's table exp(-2 pi i k / (P R)), k < P (unused when P == 1).
template <typename T, int N, int R, int P, int NT>
__device__ __forceinline__ void static_pass(cpx<T>* buf, const cpx<T>* __restrict__ tw, int tid) {
    constexpr int NB = N / R;                          // butterflies
    constexpr int B = (NB + NT - 1) / NT;              // per thread
    cpx<T> v[B][R];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = tid + b * NT;
        if (B * NT == NB || j < NB) {
            cpx<T> w1 = {(T)1, (T)0};
            if constexpr (P > 1) w1 = tw[j % P];
#pragma unroll
            for (int q = 0; q < R; ++q) v[b][q] = buf[j + q * NB];
            if constexpr (P > 1) {
                // powers of the butterfly's twiddle, depth log2: w^2 = w w, w^3 = w^2 w, w^4 = (w^2)^2, ...
                cpx<T> w[R];
                w[1] = w1;
#pragma unroll
                for (int q = 2; q < R; ++q) w[q] = cmul(w[q / 2], w[q - q / 2]);
#pragma unroll
                for (int q = 1; q < R; ++q) v[b][q] = cmul(v[b][q], w[q]);
            }
            dft_static<T, R>(v[b]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = tid + b * NT;
        if (B * NT == NB || j < NB) {
            const int k = P > 1 ? j % P : 0;
            const int base = (j - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; ++q) buf[base + q * P] = v[b][q];
        }
    }
    __syncthreads();
}

// The whole transform: radices R0 (first pass) .. ; tables of the passes concatenated (make_static_twiddles).
template <typename T, int NT, int N, int P, int R0, int... RS>
__device__ __forceinline__ void static_fft_passes(cpx<T>* buf, const cpx<T>* __restrict__ tw, int tid) {
    static_pass<T, N, R0, P, NT>(buf, tw, tid);
    if constexpr (sizeof...(RS) > 0) static_fft_passes<T, NT, N, P * R0, RS...>(buf, tw + (P > 1 ? P : 0), tid);
}

// Forward FFT of buf[0 .. N) in place, N = R0 * ...; every one of the NT threads of the workgroup calls it; ends with a barrier.
template <typename T, int NT, int R0, int... RS>
__device__ __forceinline__ void static_fft_forward(cpx<T>* buf, const cpx<T>* __restrict__ tw, int tid) {
    static_fft_passes<T, NT, (R0 * ... * RS), 1, R0, RS...>(buf, tw, tid);
}

// Host: the tables of a plan, pass by pass (the first pass has none): exp(-2 pi i k / (p R)), k < p.
template <typename T>
inline std::vector<T> make_static_twiddles(std::initializer_list<int> radices) {
    std::vector<T> t;
    const long double pi2 = 6.283185307179586476925286766559L;
    int p = 1;
    for (int R : radices) {
        if (p > 1)
            for (int k = 0; k < p; ++k) {
                const long double a = pi2 * (long double)k / ((long double)p * (long double)R);
                t.push_back((T)cosl(a));
                t.push_back((T)(-sinl(a)));
            }
        p *= R;
    }
    if (t.empty()) t.assign(2, (T)0);
    return t;
}

}  // namespace frt
