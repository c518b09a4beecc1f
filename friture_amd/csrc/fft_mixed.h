// fft_mixed.h — workgroup-wide complex FFT of any 5-smooth length, in place in LDS (gfx950).
//
// The octave bank's overlap-add stages use FFT lengths [1536, 1024, 768, 640, 576, 576, 540, 540,
// 540] (friture/filter_design.py:399-402 via _next_composite_size, friture/filter.py:250-274) and
// the delay estimator correlates windows of 24000 = 2^6 * 3 * 5^3 samples
// (friture/delay_estimator.py:114-115), so radix-2-only transforms are not enough.
//
// Shape: Stockham autosort passes of radix 4, 5, 3 or 2 over one array of n complex points in LDS.
// In a pass of radix R with p = product of the earlier radices, butterfly j (0 <= j < n/R) reads the
// points j + q*n/R, multiplies point q by exp(-2 pi i q k / (p R)) with k = j mod p, takes the R-point
// DFT and writes the results to (j - k) R + k + q p.  Reads of consecutive butterflies are consecutive
// addresses.  The pass runs in place: every thread first gathers all of its butterflies into
// registers, the workgroup meets at a barrier, then every thread scatters.
//
// Twiddles and indices (round 3).  The factor of point q of butterfly j depends on k = j mod p only: the plan carries one
// table per pass, [q - 1][k] (k < p, contiguous in k), so consecutive butterflies read consecutive entries.  Rounds 1-2
// indexed ONE table exp(-2 pi i t / n) at t = (q k step) mod n — a gather with stride q step across the lanes and two
// integer divisions by run-time operands (~30 instructions each) per point.  k itself comes from a multiply-high with
// ceil(2^32 / p) (exact for the j < 2^16, p < 2^13 of this path).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "fft_core.h"

namespace frt {

constexpr int kMaxPasses = 16;

struct MixedPlan {
    int n;
    int npass;
    int radix[kMaxPasses];
    int tw_off[kMaxPasses];          // first entry of the pass's twiddle table [R - 1][p] (make_pass_twiddles)
    unsigned magic[kMaxPasses];      // ceil(2^32 / p): j / p = umulhi(j, magic) for j < 2^16
};

// Factor n into radices {4, 5, 3, 2}; returns false when n is not 5-smooth.
inline bool make_mixed_plan(int n, MixedPlan* plan) {
    plan->n = n;
    plan->npass = 0;
    int m = n;
    auto push = [&](int r) {
        if (plan->npass < kMaxPasses) plan->radix[plan->npass] = r;
        ++plan->npass;
    };
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 5 == 0) { push(5); m /= 5; }
    while (m % 3 == 0) { push(3); m /= 3; }
    while (m % 2 == 0) { push(2); m /= 2; }
    if (!(m == 1 && plan->npass <= kMaxPasses && n >= 1 && n < (1 << 16))) return false;
    int p = 1, off = 0;
    for (int i = 0; i < plan->npass; ++i) {
        plan->tw_off[i] = off;
        plan->magic[i] = p > 1 ? (unsigned)((0x100000000ull + (unsigned long long)p - 1) / (unsigned long long)p) : 0u;
        off += (plan->radix[i] - 1) * p;
        p *= plan->radix[i];
    }
    return true;
}

// Per-pass twiddle tables of a plan, concatenated: pass i holds exp(-2 pi i q k / (p R)) at tw_off[i] + (q - 1) p + k.
template <typename T>
inline std::vector<T> make_pass_twiddles(const MixedPlan& plan) {
    std::vector<T> t;
    const long double pi2 = 6.283185307179586476925286766559L;
    int p = 1;
    for (int i = 0; i < plan.npass; ++i) {
        const int R = plan.radix[i];
        for (int q = 1; q < R; ++q)
            for (int k = 0; k < p; ++k) {
                const long double a = pi2 * (long double)q * (long double)k / ((long double)p * (long double)R);
                t.push_back((T)cosl(a));
                t.push_back((T)(-sinl(a)));
            }
        p *= R;
    }
    if (t.empty()) t.assign(2, (T)0);
    return t;
}

// Host table tw[i] = exp(-2 pi i * i / n), i < n, as interleaved (re, im) of type T.
template <typename T>
inline std::vector<T> make_twiddles(int n, int count = -1) {
    if (count < 0) count = n;
    std::vector<T> t(2 * (size_t)count);
    const long double pi2 = 6.283185307179586476925286766559L;
    for (int i = 0; i < count; ++i) {
        const long double a = pi2 * (long double)i / (long double)n;
        t[2 * i] = (T)cosl(a);
        t[2 * i + 1] = (T)(-sinl(a));
    }
    return t;
}

template <typename T>
__device__ __forceinline__ void dft3(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2) {
    const T s = (T)0.86602540378443864676;    // sin(pi/3)
    cpx<T> t = a1 + a2;
    cpx<T> d = a1 - a2;
    cpx<T> m = {a0.x - (T)0.5 * t.x, a0.y - (T)0.5 * t.y};
    cpx<T> r = {s * d.y, -s * d.x};           // -i s d
    a0 = a0 + t;
    a1 = m + r;
    a2 = m - r;
}

template <typename T>
__device__ __forceinline__ void dft5(cpx<T>& a0, cpx<T>& a1, cpx<T>& a2, cpx<T>& a3, cpx<T>& a4) {
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;    // cos(2pi/5), cos(4pi/5)
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;     // sin(2pi/5), sin(4pi/5)
    cpx<T> t1 = a1 + a4, t2 = a2 + a3, d1 = a1 - a4, d2 = a2 - a3;
    cpx<T> m1 = {a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y};
    cpx<T> m2 = {a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y};
    // -i (s1 d1 + s2 d2) and -i (s2 d1 - s1 d2)
    cpx<T> r1 = {s1 * d1.y + s2 * d2.y, -(s1 * d1.x + s2 * d2.x)};
    cpx<T> r2 = {s2 * d1.y - s1 * d2.y, -(s2 * d1.x - s1 * d2.x)};
    a0 = a0 + t1 + t2;
    a1 = m1 + r1;
    a4 = m1 - r1;
    a2 = m2 + r2;
    a3 = m2 - r2;
}

// Forward FFT of buf[0..n) in place; all `nthreads` threads of the workgroup must call it.
// MAXB >= ceil((n/2) / nthreads) butterflies per thread per pass.  tw: the plan's per-pass tables (make_pass_twiddles).
// Ends with a barrier: results are visible to the whole workgroup on return.
template <typename T, int MAXB>
__device__ void fft_mixed_forward(cpx<T>* buf, const cpx<T>* __restrict__ tw, const MixedPlan& plan, int tid, int nthreads) {
    const int n = plan.n;
    int p = 1, m = n;
    for (int pass = 0; pass < plan.npass; ++pass) {
        const int R = plan.radix[pass];
        // butterflies of the pass, n / R, without a division (R is 4, 5, 3 or 2)
        m = R == 4 ? n >> 2 : R == 2 ? n >> 1 : (int)__umulhi((unsigned)n, R == 5 ? 0x33333334u : 0x55555556u);
        const unsigned magic = plan.magic[pass];
        const cpx<T>* twp = tw + plan.tw_off[pass];
        cpx<T> v[MAXB][5];
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int j = tid + b * nthreads;
            if (j < m) {
                const int k = p > 1 ? j - p * (int)__umulhi((unsigned)j, magic) : 0;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    if (q < R) {
                        cpx<T> a = buf[j + q * m];
                        if (q > 0 && p > 1) a = cmul(a, twp[(q - 1) * p + k]);
                        v[b][q] = a;
                    }
                }
                if (R == 4) dft4(v[b][0], v[b][1], v[b][2], v[b][3]);
                else if (R == 5) dft5(v[b][0], v[b][1], v[b][2], v[b][3], v[b][4]);
                else if (R == 3) dft3(v[b][0], v[b][1], v[b][2]);
                else dft2(v[b][0], v[b][1]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int j = tid + b * nthreads;
            if (j < m) {
                const int k = p > 1 ? j - p * (int)__umulhi((unsigned)j, magic) : 0;
                const int base = (j - k) * R + k;
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    if (q < R) buf[base + q * p] = v[b][q];
            }
        }
        __syncthreads();
        p *= R;
    }
}

}  // namespace frt
