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
