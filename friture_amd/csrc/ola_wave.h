// ola_wave.h — ola_pair_kernel: the batched FFT overlap-add bank, two wavefronts per pair of windows (round 4).
//
// Reference semantics as ola_batch_kernel (ola.hip): Octave_Filters.filter fed block after block
// (friture/octavefilters.py:49-58 -> friture/filter.py:136-247) is, per stage, the running convolution of the stage input
// with 512-tap FIRs; the carried 511-sample tails are the only state.
//
// What changed against ola_batch_kernel (256 threads, one real 4096-point transform per workgroup, 250 VGPRs):
//   * two real windows of 2048 samples ride ONE complex 2048-point transform as its real and imaginary part: the filters
//     are real, so ifft(fft(a + i b) H) = (a * h) + i (b * h) — no conjugate-symmetric pack / unpack, no pairing of bin k
//     with bin N - k, no third twiddle table.  A window keeps 512 samples in front of its 1536 outputs (overlap-save);
//   * 128 threads, sixteen points each: passes of radix 16, 16, 8 inside a thread, two exchanges through one 32 KB LDS
//     array (three exchanges of radix 8 and a radix-4 tail before).  Every ds_write_b128 / ds_read_b128 lane group of the
//     exchanges covers all banks (XOR of a digit into the low address bits; no padding);
//   * the spectrum of the pair of windows stays in registers for all filters of the group, the next filter's response is
//     requested while this one's outputs are dealt with;
//   * band samples, the decimated stage signal and the block energies are produced from the registers the inverse
//     transform ends in (a thread holds runs of two consecutive outputs; consecutive threads, consecutive runs).
// (A version with ONE wavefront per pair of windows — 32 points per lane, no barrier at all — was built first and was
// correct, but its 500 registers leave one wavefront per SIMD: every LDS and table latency in the open, 1.56 ms for the
// 8-channel bank; tools/exp/README.md.)
#pragma once

#ifndef FRT_OW_TIMING            // experiment builds, = log2 of the stage length to look at: cycle counter at the phases of one
#define FRT_OW_TIMING 0          // workgroup's first filter (tools/exp/ow_timing.py)
#endif

namespace frt {

#if FRT_OW_TIMING
__device__ unsigned long long ow_timing[16];
#define OW_T(i)                                                  \
    do {                                                         \
        if (tmark) {                                             \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            if (threadIdx.x == 0) ow_timing[i] = t_;             \
        }                                                        \
    } while (0)
#else
#define OW_T(i)
#endif

constexpr int kOwN = 2048;                 // complex transform length = real window length
constexpr int kOwFront = 512;              // samples in front of a window's first output (>= 511)
constexpr int kOwL = kOwN - kOwFront;      // outputs per window
constexpr int kOwSet = 2 * kOwL;           // outputs per workgroup (window A: real part, window B: imaginary part)
constexpr int kOwThreads = 128;

// v * exp(-2 pi i K / 16)
template <int K>
__device__ __forceinline__ cpx<double> ow_tw16(cpx<double> v) {
    constexpr double h = 0.70710678118654752440, c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;
    if constexpr (K == 0) return v;
    else if constexpr (K == 4) return mul_mi(v);
    else if constexpr (K == 2) return {h * (v.x + v.y), h * (v.y - v.x)};
    else if constexpr (K == 6) return {h * (v.y - v.x), -h * (v.x + v.y)};
    else {
        // K = 1, 3, 9: (cos, -sin) of 2 pi K / 16
        constexpr double wr = K == 1 ? c1 : K == 3 ? s1 : -c1, wi = K == 1 ? -s1 : K == 3 ? -c1 : s1;
        return {v.x * wr - v.y * wi, v.x * wi + v.y * wr};
    }
}
// 16-point DFT of v[0..15] inside a thread: in v[i + 4 s], out bin q + 4 r at v[r + 4 q]
__device__ __forceinline__ void ow_dft16(cpx<double> (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dft4(v[i], v[i + 4], v[i + 8], v[i + 12]);          // over s: q at v[i + 4 q]
    v[5] = ow_tw16<1>(v[5]);
    v[6] = ow_tw16<2>(v[6]);
    v[7] = ow_tw16<3>(v[7]);
    v[9] = ow_tw16<2>(v[9]);
    v[10] = ow_tw16<4>(v[10]);
    v[11] = ow_tw16<6>(v[11]);
    v[13] = ow_tw16<3>(v[13]);
    v[14] = ow_tw16<6>(v[14]);
    v[15] = ow_tw16<9>(v[15]);
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);      // over i: r at v[r + 4 q]
}
// v[r + 4 q] *= w^(q + 4 r)
__device__ __forceinline__ void ow_twiddle16(cpx<double> (&v)[16], cpx<double> w) {
    using C = cpx<double>;
    const C w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2), w8 = cmul(w4, w4), w12 = cmul(w8, w4);
    v[4] = cmul(v[4], w);
    v[8] = cmul(v[8], w2);
    v[12] = cmul(v[12], w3);
    v[1] = cmul(v[1], w4);
    v[5] = cmul(v[5], cmul(w4, w));
    v[9] = cmul(v[9], cmul(w4, w2));
    v[13] = cmul(v[13], cmul(w4, w3));
    v[2] = cmul(v[2], w8);
    v[6] = cmul(v[6], cmul(w8, w));
    v[10] = cmul(v[10], cmul(w8, w2));
    v[14] = cmul(v[14], cmul(w8, w3));
    v[3] = cmul(v[3], w12);
    v[7] = cmul(v[7], cmul(w12, w));
    v[11] = cmul(v[11], cmul(w12, w2));
    v[15] = cmul(v[15], cmul(w12, w3));
}

// Forward transform of 2048 points held by 128 threads.  In: v[j] = z[t + 128 j].  Out: v[e + 2 d] = Z[k],
// k = e + 2 t + 256 d, e < 2, d < 8 — a thread ends with eight runs of two consecutive bins, thread t + 1 with the next runs.
//   n = n1 + 128 n2 (n1 = t), k = k2 + 16 k1:
//   pass 1  B[n1][k2] = sum_n2 z[n1 + 128 n2] W16^(n2 k2), times W2048^(n1 k2) — powers of w1 = W2048^t
//   n1 = b + 8 a (a < 16), k1 = c + 16 d:
//   pass 2  D[b][c] = sum_a (.)[b + 8 a] W16^(a c), times W128^(b c) — powers of w128 = W128^b
//   pass 3  Z[k2 + 16 (c + 16 d)] = sum_b D[b][c] W8^(b d)
// Exchange 1 moves digit a into the registers (k2 out): element address (16-byte units) b | k2 << 3 | a << 7.  Exchange 2
// moves digit b in (c out): ((k2 >> 1) ^ b) | c << 3 | (k2 & 1) << 7 | b << 8.  A write's eight-lane group differs in the low
// three address bits, a read's sixteen-lane group in the low four.  The caller's barrier discipline: nobody may still read
// xb when this is entered; the last reads of xb inside are not followed by a barrier.
__device__ __forceinline__ void ow_fft(cpx<double> (&v)[16], cpx<double>* xb, int t, cpx<double> w1, cpx<double> w128, bool tmark) {
    using C = cpx<double>;
    // addresses and twiddle powers are derived where they are used: hoisted out of the filter loop as invariants (they are)
    // they do not fit the registers, and a spilled value comes back through the vector memory pipe, in order behind the
    // table loads — a stall of a microsecond each time
    asm volatile("" : "+v"(t), "+v"(w1.x), "+v"(w1.y), "+v"(w128.x), "+v"(w128.y));
    ow_dft16(v);
    ow_twiddle16(v, w1);
    OW_T(2);
    {
        C* wr = xb + ((t & 7) + 128 * (t >> 3));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) wr[8 * (q + 4 * r)] = v[r + 4 * q];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = xb[t + 128 * j];
    }
    OW_T(3);
    ow_dft16(v);
    ow_twiddle16(v, w128);
    OW_T(4);
    asm volatile("" : "+v"(t));
    __syncthreads();                     // exchange 1 has been read by everybody
    {
        const int b = t & 7, k2h = t >> 4, k2l = (t >> 3) & 1;
        C* wr = xb + ((k2h ^ b) + 128 * k2l + 256 * b);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) wr[8 * (q + 4 * r)] = v[r + 4 * q];
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 8; ++bb)
#pragma unroll
            for (int e = 0; e < 2; ++e) v[e + 2 * bb] = xb[(t ^ bb) + 128 * (e + 2 * bb)];
    }
    OW_T(5);
    {
        C u[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = v[e + 2 * j];
            dft8(u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[e + 2 * j] = u[j];
        }
    }
}

typedef double ow_d2 __attribute__((ext_vector_type(2), aligned(8)));

// grid (sets, filter groups, channels), 128 threads.  A set = 3072 consecutive outputs of the stage's n + 511 (the outputs
// past the stage's end are the new tails).  Global addresses are a uniform base plus a 32-bit thread offset throughout.
// A = OlaBatchArgs (kernel argument) or the same struct in the constant address space (a table in device memory read with scalar
// loads): one body, two entry points below.
template <typename A>
__device__ __forceinline__ void ola_pair_body(const A& a, const int set, const int grp, const int ch, const int nsets) {
    using C = cpx<double>;
    __shared__ C xb[kOwN];
    __shared__ double wsum[2][12];
    const long long S = (long long)set * kOwSet, n = a.n;
    const bool inner = set != 0 && S + kOwSet <= n;          // uniform: every sample and every output of the set exists
    const bool tail_set = S + kOwSet > n;                    // holds the stage's end: partial runs and the new tails
    const C* tw = (const C*)a.tw;
    const C w1 = tw[threadIdx.x], w128 = tw[16 * (threadIdx.x & 7)];
    int nf = a.f_count - grp * a.gsize;
    nf = nf < a.gsize ? nf : a.gsize;
    const int fg = a.f_first + grp * a.gsize;                // first filter of this workgroup

    C v[16], xc[16];
    {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        // window A position p <-> stage sample S - 512 + p, window B: 1536 further; z[p] = A[p] + i B[p], p = t + 128 j
        if (inner) {
            if (a.x_f32) {
                const float* xs = (const float*)a.x + (long long)ch * a.x_stride + (S - kOwFront);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j].x = (double)(xs + 128 * j)[t];
#pragma unroll
                for (int j = 4; j < 16; ++j) v[j].y = (double)(xs + kOwL + 128 * j)[t];
            } else {
                const double* xs = (const double*)a.x + (long long)ch * a.x_stride + (S - kOwFront);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j].x = (xs + 128 * j)[t];
#pragma unroll
                for (int j = 4; j < 16; ++j) v[j].y = (xs + kOwL + 128 * j)[t];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j].y = v[j + 12].x;      // B's first 512 samples are A's last
        } else {
            const float* xf = (const float*)a.x + (long long)ch * a.x_stride;
            const double* xd = (const double*)a.x + (long long)ch * a.x_stride;
            const long long s0 = S - kOwFront + t;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long sa = s0 + 128 * j, sb = sa + kOwL;
                v[j].x = sa < 0 || sa >= n ? 0.0 : a.x_f32 ? (double)xf[sa] : xd[sa];
                v[j].y = sb < 0 || sb >= n ? 0.0 : a.x_f32 ? (double)xf[sb] : xd[sb];
            }
        }
        ow_fft(v, xb, t, w1, w128, false);
        asm volatile("" : "+v"(t));
        // spectrum to natural order (bin t + 128 j in register j), conjugated: the inverse is conj(fft(conj(X) conj(H) / N)),
        // conj(H) / N being the table.  Bin k sits at k ^ ((k >> 3) & 1).
        __syncthreads();
        const int K0 = 2 * (t & 7) + 16 * (t >> 3), x3 = (t >> 2) & 1;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 2; ++e) xb[K0 + (e ^ x3) + 256 * d] = v[e + 2 * d];
        __syncthreads();
        const int R3 = t ^ ((t >> 3) & 1);
        const C* H = (const C*)a.Hw + (size_t)fg * kOwN;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            xc[j] = cconj(xb[R3 + 128 * j]);
            v[j] = cmul(xc[j], (H + 128 * j)[t]);                     // the first filter's products
        }
        __syncthreads();
    }
    for (int it = 0; it < nf; ++it) {
        const int f = fg + it;
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const bool tmark = FRT_OW_TIMING && it == 0 && set == nsets / 2 && ch == 0 && grp == 0 && n == (1ll << FRT_OW_TIMING);
        OW_T(0);
        const bool dec = f == a.dec_filter;
        const bool energy = a.eblock && !dec;
        const int m = a.elen;
        const double* ewt_f = a.ewt + a.ewt_off[f];
        // this thread's block-energy weights, requested a transform ahead: a run of two outputs at block index i0 weighs
        // (y0^2 r + y1^2) w[i0 + 1], and i0 = tau mod m takes at most four values per thread (tau = 2 t + 256 (d - 2) + 1536 win)
        double wq[4] = {0.0, 0.0, 0.0, 0.0};
        if (energy && m >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wq[i] = ewt_f[((2 * t + 256 * i) & (m - 1)) + 1];
        }
        OW_T(1);
        ow_fft(v, xb, t, w1, w128, tmark);
        OW_T(6);
        asm volatile("" : "+v"(t));
        // ---- outputs of filter f: position p = e + 2 t + 256 d of window A (real part) and B (minus the imaginary part);
        // p >= 512 are outputs: set-relative index tau = p - 512 (A), p + 1024 (B), stage index S + tau
        const double* pin = a.pend_in + ((size_t)ch * a.nfilt + f) * kTail;
        double* ys = a.y && !dec ? a.y + (long long)ch * a.y_cstride + a.y_off[f] + S : nullptr;
        double* xns = a.xnext && dec ? a.xnext + (long long)ch * a.xnext_stride + S / 2 : nullptr;
        const double rr = a.er[f];
        double* eo = a.eblock + ((size_t)ch * a.nblocks) * a.nbands + a.band_index[f];
        const int nleft = (int)((n - S) < kOwSet + kTail ? (n - S) : kOwSet + kTail);      // outputs of the stage in this set (tau < nleft)
        if (tail_set) __syncthreads();                                     // xb is written below: the transform's last reads are done
        double contrib[12];
#pragma unroll
        for (int win = 0; win < 2; ++win)
#pragma unroll
            for (int d = 2; d < 8; ++d) {
                const int tau = 2 * t + 256 * (d - 2) + kOwL * win;
                double y0 = win == 0 ? v[2 * d].x : -v[2 * d].y, y1 = win == 0 ? v[2 * d + 1].x : -v[2 * d + 1].y;
                if (win == 0 && d < 4 && set == 0) {                       // the carried tails (filter.py:213-245): t < 511
                    if (tau < kTail) y0 += pin[tau];
                    if (tau + 1 < kTail) y1 += pin[tau + 1];
                }
                if (tail_set) *(ow_d2*)((double*)xb + tau) = ow_d2{y0, y1};            // kept for the pass over the stage's end below
                if (tau + 1 < nleft) {
                    if (ys) *(ow_d2*)(ys + tau) = ow_d2{y0, y1};
                    if (xns) xns[tau >> 1] = y0;
                }
                if (energy) {
                    // zero-state block energy alpha sum_i (1 - alpha)^(m-1-i) y_i^2 (exp_smoothing.py:40-56), blocks of m samples
                    double e = 0.0;
                    if (m >= 2) {
                        e = (y0 * y0 * rr + y1 * y1) * wq[(d - 2 + 6 * win) & 3];      // tau mod 1024 = 2 t + 256 ((d - 2 + 6 win) mod 4)
                    } else {                                               // m = 1: every sample is a block
                        const long long tt = S + tau;
                        if (tt < a.nblocks) eo[(size_t)tt * a.nbands] = y0 * y0 * ewt_f[0];
                        if (tt + 1 < a.nblocks) eo[(size_t)(tt + 1) * a.nbands] = y1 * y1 * ewt_f[0];
                    }
                    contrib[win * 6 + d - 2] = e;
                }
            }
        if (tail_set) {
            // elements of the runs that straddle the stage's end, and the new tails: outputs n .. n + 510 of the stage
            __syncthreads();
            const double* res = (const double*)xb;
            double* po = a.pend_out + ((size_t)ch * a.nfilt + f) * kTail;
            const int nout = nleft < 0 ? 0 : nleft < kOwSet ? nleft : kOwSet;        // tau < nout: samples of the stage
            for (int tau = (nout & ~1) + t; tau < kOwSet; tau += kOwThreads) {
                const long long tt = S + tau;
                if (tau < nout) {
                    if (ys) ys[tau] = res[tau];
                    if (xns && !(tau & 1)) xns[tau >> 1] = res[tau];
                } else if (tt - n < kTail) {
                    po[tt - n] = res[tau];
                }
            }
        }
        if (energy && m >= 2) {
            // a block's samples: m / 2 consecutive threads (m <= 256), both wavefronts' and 2 or 4 values of d (512, 1024)
            const int lane = t & 63;
            if (m <= 128) {
                const int w = m >> 1, lm = 31 - __builtin_clz(m);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const double e = group_sum(contrib[i], w);
                    const long long eg = (S + 2 * t + 256 * (i % 6) + kOwL * (i / 6)) >> lm;
                    if ((lane & (w - 1)) == 0 && eg < a.nblocks) eo[(size_t)eg * a.nbands] = e;
                }
            } else {
                // sums over a wavefront, then the two wavefronts' halves in a fixed order
                double part[12];
                int np;
                if (m == 256) {
                    np = 12;
#pragma unroll
                    for (int i = 0; i < 12; ++i) part[i] = contrib[i];
                } else if (m == 512) {
                    np = 6;
#pragma unroll
                    for (int i = 0; i < 6; ++i) part[i] = contrib[2 * i] + contrib[2 * i + 1];
                } else {                                                   // 1024: A d-2 = 0..3 | A 4, 5 + B 0, 1 | B 2..5
                    np = 3;
                    part[0] = (contrib[0] + contrib[1]) + (contrib[2] + contrib[3]);
                    part[1] = (contrib[4] + contrib[5]) + (contrib[6] + contrib[7]);
                    part[2] = (contrib[8] + contrib[9]) + (contrib[10] + contrib[11]);
                }
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    if (i < np) {
                        const double e = group_sum(part[i], 64);
                        if (lane == 0) wsum[t >> 6][i] = e;
                    }
                __syncthreads();
                if (t < np) {
                    const long long eg = (S >> (31 - __builtin_clz(m))) * 1 + t;
                    if (eg < a.nblocks) eo[(size_t)eg * a.nbands] = wsum[0][t] + wsum[1][t];
                }
            }
        }
        OW_T(7);
        __syncthreads();                                                   // xb and wsum are free again
        if (it + 1 < nf) {
            // the next filter's products.  (Its response requested earlier — before the transform, or before the outputs — does
            // not fit the 256 registers beside the spectrum: the allocator then spills the spectrum and brings it back one value
            // at a time.  Accumulation registers are no way out: a kernel that names them gets its budget split 128 / 128.)
            asm volatile("" : "+v"(t));
            const C* H = (const C*)a.Hw + (size_t)(f + 1) * kOwN;
            C hn[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) hn[j] = (H + 128 * j)[t];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = cmul(xc[j], hn[j]);
        }
    }
}

__global__ void __launch_bounds__(kOwThreads, 2) ola_pair_kernel(const OlaBatchArgs a) {
    ola_pair_body(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// Several stages' launches in one: blockIdx.x walks the sets of the stages listed in `m` one stage after the other, the stage's
// arguments come from a table in device memory (wave-uniform: scalar loads through the constant address space).
struct OlaMultiIndex {
    int nstage;
    int first_set[kNOctave + 1];        // blockIdx.x of a stage's first set; [nstage] = the grid's x
};
typedef const OlaBatchArgs __attribute__((address_space(4))) OlaBatchArgsK;
__global__ void __launch_bounds__(kOwThreads, 2) ola_pair_multi_kernel(const OlaBatchArgs* table, const OlaMultiIndex m) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < kNOctave; ++i) s += (i < m.nstage && (int)blockIdx.x >= m.first_set[i]) ? 1 : 0;
    const OlaBatchArgsK& a = ((OlaBatchArgsK*)(uintptr_t)table)[s];
    ola_pair_body(a, (int)blockIdx.x - m.first_set[s], blockIdx.y, blockIdx.z, m.first_set[s + 1] - m.first_set[s]);
}

}  // namespace frt
