// ola.hip — K3: the production octave bank of the reference, FFT overlap-add with 512-tap
// minimum-phase FIR equivalents of every IIR, for gfx950.
//
// Reference semantics: Octave_Filters.filter (friture/octavefilters.py:49-58) ->
// octave_filter_bank_decimation_fft (friture/filter.py:136-247).  Per octave stage j with FFT size
// F_j in [1536, 1024, 768, 640, 576, 576, 540, 540, 540] (friture/filter_design.py:399-402):
//     X       = rfft(x_j, F_j)                       (x_j zero padded / cropped to F_j)
//     y_f     = irfft(X * H_f, F_j)   for the bpo band-passes and the decimation low-pass
//     y_f[:a] += pending_f[:a],  a = min(511, len(x_j))
//     band output = y_f[:len(x_j)];  x_{j+1} = y_dec[:len(x_j):2]
//     pending_f  = y_f[len(x_j) : len(x_j)+511] (+ the part of the old pending not consumed yet)
//
// Kernel shape: one workgroup per (filter, channel) and one launch per stage (the stages form a
// dependency chain through the decimated signal).  A real transform of length F is a complex
// transform of length M = F/2 (fft_mixed.h: radix 4/5/3/2 Stockham passes in LDS, float64) plus
// the conjugate-symmetric pack / unpack; the spectral multiply sits between the two.  The forward
// transform of x_j is recomputed by each of the bpo+1 workgroups that need it: it is a third of
// the work of a workgroup and saves a round trip of X through HBM plus a launch.
#include <cmath>

#include "octbank.h"

namespace frt {

constexpr int kOlaThreads = 256;
constexpr int kOlaMaxM = 768;        // F <= 1536
constexpr int kOlaMaxB = 2;          // ceil((768 / 2) / 256)

struct OlaStageArgs {
    const double* x;           // [C][x_stride] stage input
    long long x_stride;
    int ns;                    // valid samples per channel
    int F, M;
    MixedPlan plan;
    const double* tw;          // [M] exp(-2 pi i t / M)
    const double* twl;         // [M+1] exp(-2 pi i k / F)
    const double* H;           // [nfilt][M+1]
    double* pending;           // [C][nfilt][kTail]
    int nfilt, dec_filter;
    double* y;                 // packed band outputs
    long long y_cstride;
    long long y_off[kMaxFilters];
    double* xnext;             // [C][xnext_stride]
    long long xnext_stride;
};

__global__ void __launch_bounds__(kOlaThreads) ola_stage_kernel(const OlaStageArgs a) {
    using C = cpx<double>;
    __shared__ C buf[kOlaMaxM];
    __shared__ C spec[kOlaMaxM + 1];

    const int tid = threadIdx.x;
    const int f = blockIdx.x, c = blockIdx.y;
    const int M = a.M, F = a.F, ns = a.ns;
    const C* tw = (const C*)a.tw;
    const C* twl = (const C*)a.twl;
    const C* H = (const C*)a.H + (size_t)f * (M + 1);
    const double* x = a.x + (long long)c * a.x_stride;

    // z[n] = x[2n] + i x[2n+1], zero padded (rfft(x, F) also crops to F samples)
    for (int n = tid; n < M; n += kOlaThreads) {
        const int t = 2 * n;
        buf[n] = {t < ns ? x[t] : 0.0, t + 1 < ns ? x[t + 1] : 0.0};
    }
    __syncthreads();
    fft_mixed_forward<double, kOlaMaxB>(buf, tw, a.plan, tid, kOlaThreads);

    // unpack to X[k], k = 0..M, and multiply by the filter response
    for (int k = tid; k <= M; k += kOlaThreads) {
        const C A = buf[k == M ? 0 : k];
        const C B = cconj(buf[k == 0 ? 0 : M - k]);
        const C S = A + B, D = A - B;
        const C t = cmul(twl[k], D);
        const C X = {0.5 * (S.x + t.y), 0.5 * (S.y - t.x)};
        C Y = cmul(X, H[k]);
        if (k == 0 || k == M) Y.y = 0.0;          // irfft ignores the imaginary part of the edge bins
        spec[k] = Y;
    }
    __syncthreads();
    // pack for the inverse: Z[k] = ((A + B) + i conj(w^k) (A - B)) / 2, stored conjugated
    for (int k = tid; k < M; k += kOlaThreads) {
        const C A = spec[k];
        const C B = cconj(spec[M - k]);
        const C S = A + B, D = A - B;
        const C t = cmul(cconj(twl[k]), D);
        buf[k] = {0.5 * (S.x - t.y), -0.5 * (S.y + t.x)};
    }
    __syncthreads();
    fft_mixed_forward<double, kOlaMaxB>(buf, tw, a.plan, tid, kOlaThreads);   // conj(FFT(conj Z)) = M * ifft(Z)

    const double inv = 1.0 / (double)M;
    auto full = [&](int t) -> double {            // y_full[t], t < F
        const C v = buf[t >> 1];
        return (t & 1) ? -v.y * inv : v.x * inv;
    };
    double* pend = a.pending + ((size_t)c * a.nfilt + f) * kTail;
    const int add = ns < kTail ? ns : kTail;

    // outputs (reads of the old pending tail first, writes of the new one after the barrier)
    if (f == a.dec_filter) {
        if (a.xnext) {
            double* xn = a.xnext + (long long)c * a.xnext_stride;
            for (int m = tid; 2 * m < ns; m += kOlaThreads) {
                const int t = 2 * m;
                xn[m] = t < add ? full(t) + pend[t] : full(t);
            }
        }
    } else {
        double* y = a.y + (long long)c * a.y_cstride + a.y_off[f];
        for (int t = tid; t < ns; t += kOlaThreads) y[t] = t < add ? full(t) + pend[t] : full(t);
    }
    double tail[(kTail + kOlaThreads - 1) / kOlaThreads];
#pragma unroll
    for (int b = 0; b < (kTail + kOlaThreads - 1) / kOlaThreads; ++b) {
        const int t = tid + b * kOlaThreads;
        double v = 0.0;
        if (t < kTail) {
            v = ns + t < F ? full(ns + t) : 0.0;
            if (add + t < kTail) v += pend[add + t];
        }
        tail[b] = v;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < (kTail + kOlaThreads - 1) / kOlaThreads; ++b) {
        const int t = tid + b * kOlaThreads;
        if (t < kTail) pend[t] = tail[b];
    }
}

}  // namespace frt

using namespace frt;

static int next_smooth_size(int n) {   // friture/filter.py:250-274
    int p2 = 1;
    while (p2 < n) p2 *= 2;
    for (int size = n; size < p2; ++size) {
        int s = size;
        for (int p : {2, 3, 5})
            while (s % p == 0) s /= p;
        if (s == 1) return size;
    }
    return p2;
}

int frt_ola_create(frt_octbank* h, const double* boct_fir, const double* bdec_fir) {
    frt_ola_state* o = new frt_ola_state();
    h->ola = o;
    const int nfilt = h->nfilt;
    int rc;
    o->h_taps.resize((size_t)nfilt * kFirLength);
    for (int f = 0; f < nfilt; ++f)
        memcpy(&o->h_taps[(size_t)f * kFirLength], f < h->bpo ? boct_fir + (size_t)f * kFirLength : bdec_fir, kFirLength * sizeof(double));
    for (int j = 0; j < kNOctave; ++j) {
        const int F = next_smooth_size((1024 >> j) + kFirLength - 1);     // filter_design.py:399-402
        const int M = F / 2;
        o->fft_size[j] = F;
        FRT_REQUIRE(F % 2 == 0 && M <= kOlaMaxM && make_mixed_plan(M, &o->plan[j]), "frt_ola_create: bad FFT size %d", F);
        if ((rc = upload(o->tw[j], make_twiddles<double>(M))) || (rc = upload(o->twl[j], make_twiddles<double>(F, M + 1)))) return rc;
        // H_f[k] = sum_t h_f[t] exp(-2 pi i k t / F): the rfft of the zero-padded taps
        std::vector<long double> ct(F), st(F);
        const long double pi2 = 6.283185307179586476925286766559L;
        for (int t = 0; t < F; ++t) {
            ct[t] = cosl(pi2 * t / F);
            st[t] = sinl(pi2 * t / F);
        }
        std::vector<double> Hh((size_t)nfilt * (M + 1) * 2);
        for (int f = 0; f < nfilt; ++f) {
            const double* taps = f < h->bpo ? boct_fir + (size_t)f * kFirLength : bdec_fir;
            for (int k = 0; k <= M; ++k) {
                long double re = 0, im = 0;
                for (int t = 0; t < kFirLength; ++t) {
                    const int idx = (int)(((long long)k * t) % F);
                    re += taps[t] * ct[idx];
                    im -= taps[t] * st[idx];
                }
                Hh[((size_t)f * (M + 1) + k) * 2] = (double)re;
                Hh[((size_t)f * (M + 1) + k) * 2 + 1] = (double)im;
            }
        }
        if ((rc = upload(o->H[j], Hh))) return rc;
    }
    const size_t pbytes = (size_t)kNOctave * h->n_channels * nfilt * kTail * sizeof(double);
    if ((rc = o->pending.reserve(pbytes))) return rc;
    FRT_HIP_CHECK(hipMemset(o->pending.ptr, 0, pbytes));
    return FRT_OK;
}

void frt_ola_destroy(frt_octbank* h) {
    if (!h || !h->ola) return;
    for (int j = 0; j < kNOctave; ++j) {
        h->ola->tw[j].release();
        h->ola->twl[j].release();
        h->ola->H[j].release();
    }
    h->ola->pending.release();
    h->ola->pending_next.release();
    h->ola->btw.release();
    h->ola->btwl.release();
    h->ola->bH.release();
    h->ola->ewt.release();
    delete h->ola;
    h->ola = nullptr;
}

int frt_ola_reset(frt_octbank* h) {
    FRT_HIP_CHECK(hipMemsetAsync(h->ola->pending.ptr, 0, h->ola->pending.bytes, h->stream));
    return FRT_OK;
}

int frt_ola_filter(frt_octbank* h, const double* d_x, int n, double* d_y, int64_t y_cstride) {
    frt_ola_state* o = h->ola;
    int len[kNOctave];
    stage_lengths(n, len);
    int rc;
    for (int j = 1; j < kNOctave; ++j)
        if ((rc = h->xbuf[j].reserve((size_t)h->n_channels * len[j] * sizeof(double)))) return rc;
    std::vector<long long> band_off(h->nbands + 1, 0);
    for (int k = 0; k < h->nbands; ++k) band_off[k + 1] = band_off[k] + len[kNOctave - 1 - k / h->bpo];
    for (int j = 0; j < kNOctave; ++j) {
        OlaStageArgs a{};
        a.x = j == 0 ? d_x : h->xbuf[j].as<double>();
        a.x_stride = len[j];
        a.ns = len[j];
        a.F = o->fft_size[j];
        a.M = a.F / 2;
        a.plan = o->plan[j];
        a.tw = o->tw[j].as<double>();
        a.twl = o->twl[j].as<double>();
        a.H = o->H[j].as<double>();
        a.pending = o->pending.as<double>() + (size_t)j * h->n_channels * h->nfilt * kTail;
        a.nfilt = h->nfilt;
        a.dec_filter = h->bpo;
        a.y = d_y;
        a.y_cstride = y_cstride;
        for (int i = 0; i < h->bpo; ++i) a.y_off[i] = band_off[(kNOctave - 1 - j) * h->bpo + i];
        a.xnext = j + 1 < kNOctave ? h->xbuf[j + 1].as<double>() : nullptr;
        a.xnext_stride = j + 1 < kNOctave ? len[j + 1] : 0;
        hipLaunchKernelGGL(ola_stage_kernel, dim3(h->nfilt, h->n_channels), dim3(kOlaThreads), 0, h->stream, a);
        FRT_HIP_CHECK(hipGetLastError());
    }
    return FRT_OK;
}


// ---- batched overlap-add bank ---------------------------------------------------------------------------------------
// Octave_Filters.filter fed block after block (octavefilters.py:49-58 -> filter.py:136-247) IS, per stage, the running
// convolution of the stage input with a 512-tap FIR: the per-block overlap-add (pending tails added into the next blocks,
// filter.py:213-245) only fixes the order in which the same products are summed, and every block of the 1024-sample
// cadence has an even length at every stage, so the per-block decimation y[:N_s:2] picks the even samples of the whole
// stream.  Nothing in a stage depends on an earlier block of the SAME stage except through those 511-sample tails — no
// recurrence — so a stage of a long batch is one launch over (block, filter group, channel):
//   * a workgroup takes kObL = 3072 output samples of the stage and reads their 3072 + 511 input samples (overlap-save:
//     the 511 samples in front replace the neighbour's tail; in front of the batch they are zeros and the carried tails
//     `pend_in` are added to the first 511 outputs instead, exactly the reference's state),
//   * ONE forward real FFT of length 4096 (complex 2048: the STFT's radix-8 Stockham engine, fft_core.h, eight points
//     per thread, float64), kept in LDS as X[0..2048],
//   * per filter of its group: Y = X H_f, inverse, then band output / decimated stage output / block energies straight
//     from LDS; the workgroup holding the end of the stage also writes the new tails (its window ends in 511 + zeros).
// 4096 = 3072 + 511 + 511 + 2: the circular convolution never wraps into a sample that is used.
constexpr int kObF = 4096, kObM = kObF / 2, kObL = 3072, kObThreads = 256;

struct OlaBatchArgs {
    const void* x;             // [C][x_stride] stage input: float (x_f32) or double
    int x_f32;
    long long x_stride;
    long long n;               // samples of this stage per channel
    const double* tw;          // [M] exp(-2 pi i t / M)
    const double* twl;         // [M+1] exp(-2 pi i k / F)
    const double* H;           // [nfilt][M+1]
    const double* pend_in;     // [C][nfilt][kTail]
    double* pend_out;
    int nfilt, dec_filter, gsize;
    double* y;                 // packed band outputs or null
    long long y_cstride;
    long long y_off[kMaxFilters];
    double* xnext;             // [C][xnext_stride] or null
    long long xnext_stride;
    double* eblock;            // [C][nblocks][nbands] or null
    int elen;                  // samples of this stage per energy block (power of two, divides kObL)
    int nblocks, nbands;
    int band_index[kMaxFilters];
    const double* ewt;         // smoothing weights alpha (1 - alpha)^(elen - 1 - i), per band at ewt_off
    long long ewt_off[kMaxFilters];
};

__global__ void __launch_bounds__(kObThreads) ola_batch_kernel(const OlaBatchArgs a) {
    using C = cpx<double>;
    constexpr int M = kObM, F = kObF, LOG2M = 11;
    using P = Pow2Plan<LOG2M>;
    static_assert(P::M == M && P::TPF == kObThreads, "one thread per eight points of the 2048-point complex transform");
    __shared__ C buf[lds_padded_size(M)];         // exchange buffer of the radix-8 passes; afterwards Z, then the finished window
    __shared__ C spec[M + 1];                     // X[0..M]
    const int tid = threadIdx.x;
    const int blk = blockIdx.x, grp = blockIdx.y, c = blockIdx.z;
    const long long o0 = (long long)blk * kObL;                  // first output sample of this workgroup
    const int Lb = (int)((a.n - o0) < kObL ? (a.n - o0) : kObL); // its outputs
    const bool first = blk == 0, last = o0 + Lb == a.n;
    const C* twl = (const C*)a.twl;
    const TwTable<double, LOG2M> twt{(const C*)a.tw, 0};

    // window position p <-> stage sample o0 - 511 + p; z[q] = w[2q] + i w[2q+1]; thread tid holds q = tid + 256 j
    C v[8];
    {
        const long long s0 = o0 - kTail;
        const float* xf = (const float*)a.x + (long long)c * a.x_stride;
        const double* xd = (const double*)a.x + (long long)c * a.x_stride;
        auto sample = [&](long long s) -> double {
            if (s < 0 || s >= a.n) return 0.0;
            return a.x_f32 ? (double)xf[s] : xd[s];
        };
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = tid + j * kObThreads;
            v[j] = {sample(s0 + 2 * q), sample(s0 + 2 * q + 1)};
        }
    }
    fft_pow2_forward<double, LOG2M, false>(v, buf, tid, twt);    // v[j] = Z[tid + 256 j]
    __syncthreads();                                             // the last pass's gathers are done: buf is free
#pragma unroll
    for (int j = 0; j < 8; ++j) buf[tid + j * kObThreads] = v[j];
    __syncthreads();
    for (int k = tid; k <= M; k += kObThreads) {                 // X[k], k = 0..M
        const C A = buf[k == M ? 0 : k];
        const C B = cconj(buf[k == 0 ? 0 : M - k]);
        const C S = A + B, D = A - B;
        const C t = cmul(twl[k], D);
        spec[k] = {0.5 * (S.x + t.y), 0.5 * (S.y - t.x)};
    }
    __syncthreads();

    double* out = (double*)buf;                                  // the finished window, plain doubles, after each inverse
    const double inv = 1.0 / (double)M;
    for (int fi = 0; fi < a.gsize; ++fi) {
        const int f = grp * a.gsize + fi;
        if (f >= a.nfilt) break;
        const C* H = (const C*)a.H + (size_t)f * (M + 1);
        // Y = X H packed for the inverse: Z[k] = ((A + B) + i conj(w^k) (A - B)) / 2 with A = Y[k], B = conj Y[M-k], conjugated
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = tid + j * kObThreads;
            C A = cmul(spec[k], H[k]);
            C Bm = cmul(spec[M - k], H[M - k]);
            if (k == 0) { A.y = 0.0; Bm.y = 0.0; }              // irfft ignores the imaginary part of the edge bins
            const C B = cconj(Bm);
            const C S = A + B, D = A - B;
            const C t = cmul(cconj(twl[k]), D);
            v[j] = {0.5 * (S.x - t.y), -0.5 * (S.y + t.x)};
        }
        fft_pow2_forward<double, LOG2M, false>(v, buf, tid, twt);               // conj(FFT(conj Z)) = M ifft(Z)
        __syncthreads();
        const double* pin = a.pend_in + ((size_t)c * a.nfilt + f) * kTail;
        // finish the window: scale / sign, carried tails on the first 511 outputs of the batch; plain doubles in LDS
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = tid + j * kObThreads;
            double e = v[j].x * inv, o = -v[j].y * inv;
            if (first) {
                const int t0 = 2 * q - kTail;                    // output index of the even slot
                if (t0 >= 0 && t0 < kTail && t0 < Lb) e += pin[t0];
                if (t0 + 1 >= 0 && t0 + 1 < kTail && t0 + 1 < Lb) o += pin[t0 + 1];
            }
            out[2 * q] = e;
            out[2 * q + 1] = o;
        }
        __syncthreads();
        const double* res = out + kTail;                         // res[t]: output o0 + t, t < Lb (t >= Lb: the tail)
        if (f == a.dec_filter) {
            if (a.xnext) {
                double* xn = a.xnext + (long long)c * a.xnext_stride + o0 / 2;      // o0 is even
                for (int m = tid; 2 * m < Lb; m += kObThreads) xn[m] = res[2 * m];
            }
        } else {
            if (a.y) {
                double* y = a.y + (long long)c * a.y_cstride + a.y_off[f] + o0;
                for (int t = tid; t < Lb; t += kObThreads) y[t] = res[t];
            }
            if (a.eblock) {
                // zero-state block energies alpha sum_i (1-alpha)^(m-1-i) y_i^2 (exp_smoothing.py:40-56): groups of
                // w = min(m, 64) lanes per energy block, fixed summation order
                // (a power of two below 64: that many lanes per block and several blocks per wave; anything else: a whole wave)
                const int m = a.elen, w = (m < 64 && (m & (m - 1)) == 0) ? m : 64, per_wave = 64 / w;
                const int lane = tid & 63, wave = tid >> 6, sub = lane / w, li = lane - sub * w;
                const double* wt = a.ewt + a.ewt_off[f];
                const int ne = Lb / m;
                double* eo = a.eblock + ((size_t)c * a.nblocks + (size_t)(o0 / m)) * a.nbands + a.band_index[f];
                for (int e0 = wave * per_wave; e0 < ne; e0 += (kObThreads / 64) * per_wave) {
                    const int le = e0 + sub;
                    double acc = 0.0;
                    if (le < ne)
                        for (int i2 = li; i2 < m; i2 += w) {
                            const double val = res[le * m + i2];
                            acc += wt[i2] * (val * val);
                        }
                    for (int d = w >> 1; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
                    if (le < ne && li == 0) eo[(size_t)le * a.nbands] = acc;
                }
            }
        }
        if (last) {
            double* po = a.pend_out + ((size_t)c * a.nfilt + f) * kTail;
            for (int t = tid; t < kTail; t += kObThreads) {
                double val = res[Lb + t];                        // 511 + Lb + 510 < 4096
                if (first && a.n + t < kTail) val += pin[a.n + t];      // a batch shorter than the tails it inherited
                po[t] = val;
            }
        }
        __syncthreads();
    }
}

static int ola_batch_tables(frt_octbank* h) {
    frt_ola_state* o = h->ola;
    if (o->bH.ptr) return FRT_OK;
    int rc;
    if ((rc = upload(o->btw, make_twiddles<double>(kObM))) || (rc = upload(o->btwl, make_twiddles<double>(kObF, kObM + 1)))) return rc;
    // H_f[k] = sum_t h_f[t] exp(-2 pi i k t / F): the rfft of the zero-padded taps (filter_design.py computes the same
    // with numpy at the reference's own sizes)
    const int nfilt = h->nfilt;
    std::vector<long double> ct(kObF), st(kObF);
    const long double pi2 = 6.283185307179586476925286766559L;
    for (int t = 0; t < kObF; ++t) {
        ct[t] = cosl(pi2 * t / kObF);
        st[t] = sinl(pi2 * t / kObF);
    }
    std::vector<double> Hh((size_t)nfilt * (kObM + 1) * 2);
    for (int f = 0; f < nfilt; ++f) {
        const double* taps = &o->h_taps[(size_t)f * kFirLength];
        for (int k = 0; k <= kObM; ++k) {
            long double re = 0, im = 0;
            for (int t = 0; t < kFirLength; ++t) {
                const int idx = (k * t) & (kObF - 1);
                re += taps[t] * ct[idx];
                im -= taps[t] * st[idx];
            }
            Hh[((size_t)f * (kObM + 1) + k) * 2] = (double)re;
            Hh[((size_t)f * (kObM + 1) + k) * 2 + 1] = (double)im;
        }
    }
    if ((rc = upload(o->bH, Hh))) return rc;
    if ((rc = o->pending_next.reserve(o->pending.bytes))) return rc;
    return FRT_OK;
}

int frt_ola_filter_batch(frt_octbank* h, const void* d_x, int x_f32, int64_t n, double* d_y, int64_t y_cstride,
                         double* d_eblock, int eblock0, int nblocks, const double* alphas) {
    frt_ola_state* o = h->ola;
    // one energy block = the whole call (the widget's chunk, any length): a band's block is its stage's whole output
    const bool whole = d_eblock && nblocks == 1 && eblock0 == n;
    int rc;
    if ((rc = ola_batch_tables(h))) return rc;
    long long len[kNOctave];
    len[0] = n;
    for (int j = 1; j < kNOctave; ++j) len[j] = (len[j - 1] + 1) / 2;
    for (int j = 1; j < kNOctave; ++j)
        if ((rc = h->xbuf[j].reserve((size_t)h->n_channels * len[j] * sizeof(double)))) return rc;
    std::vector<long long> band_off(h->nbands + 1, 0);
    for (int k = 0; k < h->nbands; ++k) band_off[k + 1] = band_off[k] + len[kNOctave - 1 - k / h->bpo];
    if (d_eblock) {
        // smoothing weights per band: alpha (1 - alpha)^(m - 1 - i), m = eblock0 / dec (exp_smoothing.py:40-56 with the
        // kernels of octavespectrum.py:77-81)
        bool same = o->ewt_block == eblock0 && (int)o->ewt_alpha.size() == h->nbands;
        for (int k = 0; same && k < h->nbands; ++k) same = o->ewt_alpha[k] == alphas[k];
        if (!same) {
            o->ewt_off.assign(h->nbands, 0);
            std::vector<double> wt;
            long long slen[kNOctave];
            slen[0] = n;
            for (int j = 1; j < kNOctave; ++j) slen[j] = (slen[j - 1] + 1) / 2;
            for (int k = 0; k < h->nbands; ++k) {
                const int m = whole ? (int)slen[kNOctave - 1 - k / h->bpo] : eblock0 >> (kNOctave - 1 - k / h->bpo);
                o->ewt_off[k] = (long long)wt.size();
                for (int i = 0; i < m; ++i) wt.push_back(alphas[k] * std::pow(1.0 - alphas[k], (double)(m - 1 - i)));
            }
            FRT_HIP_CHECK(hipStreamSynchronize(h->stream));      // the old table may still be read
            if ((rc = upload(o->ewt, wt))) return rc;
            o->ewt_block = eblock0;
            o->ewt_alpha.assign(alphas, alphas + h->nbands);
        }
    }
    const size_t stage_pend = (size_t)h->n_channels * h->nfilt * kTail;
    for (int j = 0; j < kNOctave; ++j) {
        OlaBatchArgs a{};
        a.x = j == 0 ? d_x : h->xbuf[j].ptr;
        a.x_f32 = j == 0 ? x_f32 : 0;
        a.x_stride = len[j];
        a.n = len[j];
        a.tw = o->btw.as<double>();
        a.twl = o->btwl.as<double>();
        a.H = o->bH.as<double>();
        a.pend_in = o->pending.as<double>() + (size_t)j * stage_pend;
        a.pend_out = o->pending_next.as<double>() + (size_t)j * stage_pend;
        a.nfilt = h->nfilt;
        a.dec_filter = h->bpo;
        a.y = d_y;
        a.y_cstride = y_cstride;
        a.eblock = d_eblock;
        a.elen = !d_eblock ? 1 : whole ? (int)len[j] : (eblock0 >> j);
        a.nblocks = nblocks;
        a.nbands = h->nbands;
        a.ewt = o->ewt.as<double>();
        for (int i = 0; i < h->bpo; ++i) {
            const int band = (kNOctave - 1 - j) * h->bpo + i;
            a.y_off[i] = band_off[band];
            a.band_index[i] = band;
            a.ewt_off[i] = d_eblock ? o->ewt_off[band] : 0;
        }
        a.xnext = j + 1 < kNOctave ? h->xbuf[j + 1].as<double>() : nullptr;
        a.xnext_stride = j + 1 < kNOctave ? len[j + 1] : 0;
        const long long nblk = (len[j] + kObL - 1) / kObL;
        // filter groups: every workgroup repeats the forward transform of its window, so as few groups as still fill the chip
        int groups = 1;
        while (groups < h->nfilt && nblk * h->n_channels * groups < 2ll * device_cu_count()) ++groups;
        a.gsize = (h->nfilt + groups - 1) / groups;
        groups = (h->nfilt + a.gsize - 1) / a.gsize;
        hipLaunchKernelGGL(ola_batch_kernel, dim3((unsigned)nblk, groups, h->n_channels), dim3(kObThreads), 0, h->stream, a);
        FRT_HIP_CHECK(hipGetLastError());
    }
    std::swap(o->pending.ptr, o->pending_next.ptr);             // equal sizes; the streaming path and the graphs follow `pending`
    return FRT_OK;
}
