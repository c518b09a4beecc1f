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
        if ((rc = upload(o->tw[j], make_pass_twiddles<double>(o->plan[j]))) || (rc = upload(o->twl[j], make_twiddles<double>(F, M + 1)))) return rc;
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
    h->ola->bHw.release();
    h->ola->ewt.release();
    h->ola->taps.release();
    h->ola->ewt_off_dev.release();
    h->ola->xs.release();
    for (int p = 0; p < 2; ++p) {
        h->ola->multi_tab[p].release();
        if (h->ola->multi_pin[p]) (void)hipHostFree(h->ola->multi_pin[p]);
    }
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
[[maybe_unused]] constexpr int kObF = 4096, kObM = kObF / 2, kObL = 3072, kObThreads = 256;      // (ola_batch_kernel: -DFRT_EXPERIMENTS builds)

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
    int f_first, f_count;      // ola_pair_kernel: the launch serves filters f_first .. f_first + f_count - 1 in groups of gsize
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
    double ewr[kMaxFilters];   // (1 - alpha)^w of the band, w = lanes per energy block (see the kernel)
    const double* Hw;          // ola_pair_kernel: [nfilt][2048] conj(H_f) / 2048 of the 2048-point transform
    double er[kMaxFilters];    // ola_pair_kernel: 1 - alpha of the band
};

#ifndef FRT_OB_ABLATE           // experiment builds only (wrong results): 1 twiddle gathers -> one address, 2 H loads -> one address,
#define FRT_OB_ABLATE 0         // 4 no band outputs / energies / tails, 8 no workgroup barriers inside the transforms
#endif
struct ObTwOneAddress {         // FRT_OB_ABLATE & 1
    const cpx<double>* tw;
    __device__ __forceinline__ cpx<double> get(int n, int /*idx*/) const { return tw[n]; }
};
// fft_pow2_forward asks for its factors in order: pass 1 q = 1..7, pass 2 q = 1..7, radix-4 pass c = 0: q = 1..3, c = 1: q = 1..3
struct ObTwPowers {
    cpx<double> w1[4];
    mutable cpx<double> cur;
    __device__ __forceinline__ void load(const cpx<double>* tw, int i) {
        w1[0] = tw[((i & 7) * (kObM / 64)) & (kObM - 1)];
        w1[1] = tw[((i & 63) * (kObM / 512)) & (kObM - 1)];
        w1[2] = tw[i & 511];
        w1[3] = tw[(i + kObThreads) & 511];
    }
    __device__ __forceinline__ cpx<double> get(int n, int /*idx*/) const {
        const int row = n < 7 ? 0 : n < 14 ? 1 : n < 17 ? 2 : 3;
        const int q = n < 7 ? n : n < 14 ? n - 7 : n < 17 ? n - 14 : n - 17;      // 0 = first power
        cur = q == 0 ? w1[row] : cmul(cur, w1[row]);
        return cur;
    }
};
// Sum over the w = 2^n <= 64 lanes of a group, every lane of the group receiving it: the first four steps (partners inside a row
// of 16 lanes) are DPP moves — quad permutes, then the half-row and row mirrors, which pair a lane with one that already
// holds the other half's sum — and only the steps across rows go through ds_bpermute.  (Six bpermute round trips per block
// and band were a visible part of the bank's time.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum(double acc, int w) {
    if (w >= 2) acc += dpp_f64<0xB1>(acc);         // quad_perm [1,0,3,2]
    if (w >= 4) acc += dpp_f64<0x4E>(acc);         // quad_perm [2,3,0,1]
    if (w >= 8) acc += dpp_f64<0x141>(acc);        // row_half_mirror
    if (w >= 16) acc += dpp_f64<0x140>(acc);       // row_mirror
    if (w >= 32) acc += __shfl_xor(acc, 16, 64);
    if (w >= 64) acc += __shfl_xor(acc, 32, 64);
    return acc;
}
#ifdef FRT_EXPERIMENTS          // round 2's kernel (one real 4096-point transform per workgroup): superseded by ola_pair_kernel
                                // (ola_wave.h), kept for the A/B builds of tools/exp (FRT_OLA_NO_WAVE)
// The 2048-point forward transform of fft_core.h (radix 8, 8, 8, 4; v[j] = z[i + 256 j] in, Z[i + 256 j] out) with its three
// exchanges alternating between two LDS buffers, first -> second -> first: the barrier that would keep a pass's scatter from
// overtaking the previous pass's gathers is not needed when the scatter goes to the other buffer (whose last readers all
// passed the barrier in between).  `first` must have no readers left at entry, `second` none by the first barrier inside.
template <typename TW>
__device__ __forceinline__ void fft2048_two_buffers(cpx<double> (&v)[8], cpx<double>* first, cpx<double>* second, int i, const TW& tw) {
    using P = Pow2Plan<11>;
    constexpr int TPF = P::TPF, M = P::M;
    static_assert(P::NP8 == 3 && P::RLAST == 4, "8 x 8 x 8 x 4");
    int n = 0, p = 1;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int k = i & (p - 1);
        if (pass > 0) {
            const int step = M / (8 * p);
#pragma unroll
            for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw.get(n++, (q * k * step) & (M - 1)));
        }
        dft8(v);
        cpx<double>* buf = (pass & 1) ? second : first;
        const int base = (i - k) * 8 + k;
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[lds_pad(base + q * p)] = v[q];
        if (!(FRT_OB_ABLATE & 8)) __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = buf[lds_pad(i + j * TPF)];
        p *= 8;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {                                // two radix-4 butterflies on slots c + 2q
        const int k = (i + c * TPF) & (p - 1);
#pragma unroll
        for (int q = 1; q < 4; ++q) v[c + 2 * q] = cmul(v[c + 2 * q], tw.get(n++, (q * k * (M / (4 * p))) & (M - 1)));
        dft4(v[c], v[c + 2], v[c + 4], v[c + 6]);
    }
}
#ifndef FRT_OB_EABL     // experiment builds (wrong energies): 1 no stores, 2 no lane reduction, 4 no LDS reads, 8 no weight recurrence
#define FRT_OB_EABL 0
#endif
#ifndef FRT_OB_MIN_WAVES        // waves per SIMD the register budget is capped for (2: 256 VGPRs; occupancy is not what binds this kernel)
#define FRT_OB_MIN_WAVES 2
#endif
__global__ void __launch_bounds__(kObThreads, FRT_OB_MIN_WAVES) ola_batch_kernel(const OlaBatchArgs a) {
    using C = cpx<double>;
    constexpr int M = kObM, F = kObF, LOG2M = 11;
    using P = Pow2Plan<LOG2M>;
    static_assert(P::M == M && P::TPF == kObThreads, "one thread per eight points of the 2048-point complex transform");
    // TWO LDS arrays taking turns as exchange buffer of the radix-8 passes, hand-over of Z / of the packed products between
    // threads, and finished window: writing to the array nobody can still be reading saves the five barriers per transform
    // that only guarded the re-use of a single one (the register budget allows two workgroups per CU; two arrays of 37 KB
    // each still fit twice).  The spectrum X of the window lives in registers — thread tid keeps the four bin pairs
    // (k, M - k), k = tid + 256 j (thread 0 also bin M/2); a pair shares its sum / difference / twiddle product between the
    // two bins, in both directions.
    __shared__ C buf_a[lds_padded_size(M)];       // >= M + 1 elements each
    __shared__ C buf_b[lds_padded_size(M)];
    __shared__ C xmid_lds;                        // X[M/2] (self-paired; thread 0 packs it)
    const int tid = threadIdx.x;
    const int blk = blockIdx.x, grp = blockIdx.y, c = blockIdx.z;
    const long long o0 = (long long)blk * kObL;                  // first output sample of this workgroup
    const int Lb = (int)((a.n - o0) < kObL ? (a.n - o0) : kObL); // its outputs
    const bool first = blk == 0, last = o0 + Lb == a.n;
    const C* twl = (const C*)a.twl;
#if FRT_OB_ABLATE & 1
    const ObTwOneAddress twt{(const C*)a.tw};
#else
    // The transform's twiddle factors: per pass the thread keeps ONE factor in registers (exp(-2 pi i k step / M) of its
    // k: four complex numbers for the three twiddled passes, the radix-4 pass has two butterflies) and raises it to the
    // powers q = 2..7 (2, 3) as the pass consumes them.  Re-read from the table at every use the 20 factors were gathers
    // with a stride of q k step entries (up to 64 cache lines per wave-instruction), each pass waiting for its own L2
    // round trip: 30 % of the whole bank's time; all 20 in registers (80 VGPRs) spill.
    ObTwPowers twt;
    twt.load((const C*)a.tw, tid);
#endif

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
    fft2048_two_buffers(v, buf_a, buf_b, tid, twt);              // v[j] = Z[tid + 256 j]; exchanges in a, b, a
#pragma unroll
    for (int j = 0; j < 8; ++j) buf_b[tid + j * kObThreads] = v[j];
    __syncthreads();
    C* buf = buf_b;                                              // the array just written; `other`: free of readers
    C* other = buf_a;
    // X[k] = (S - i t) / 2 and X[M - k] = (conj S - i conj t) / 2 with S = Z[k] + conj Z[M-k], t = w^k (Z[k] - conj Z[M-k]),
    // w = exp(-2 pi i / F); k = 0 pairs X[0] with the Nyquist bin X[M]
    C xa[4], xb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = tid + j * kObThreads;                      // 0 .. M/2 - 1
        const C A = buf[k];
        const C B = cconj(buf[(M - k) & (M - 1)]);
        const C S = A + B, D = A - B;
        const C t = cmul(twl[k], D);
        xa[j] = {0.5 * (S.x + t.y), 0.5 * (S.y - t.x)};
        xb[j] = {0.5 * (S.x - t.y), -0.5 * (S.y + t.x)};
    }
    if (tid == 0) {
        const C A = buf[M / 2];                                  // self-paired: B = conj A, w^(M/2) = -i
        xmid_lds = {A.x, -A.y};                                    // (S - i (-i)(D)) / 2 with S = (2 A.x, 0), D = (0, 2 A.y)
    }
    // (no barrier: the first filter packs into `other`; `buf` is written again only behind the barrier after the packing)

    const double inv = 1.0 / (double)M;
    for (int fi = 0; fi < a.gsize; ++fi) {
        const int f = grp * a.gsize + fi;
        if (f >= a.nfilt) break;
        const C* H = (const C*)a.H + (size_t)f * (M + 1);
        // Block energies (below) weight sample i of a block of m by alpha (1 - alpha)^(m-1-i); lane li of a block's group of w
        // lanes takes the samples li + u w.  It reads ONE weight from the table — that of its last sample, requested here, a
        // whole transform ahead of its use — and steps down from it with (1 - alpha)^w (from the largest weight down, so
        // that an underflow happens where the table's own entries underflow).
        const bool band_energy = a.eblock && f != a.dec_filter && !(FRT_OB_ABLATE & 4);
        const int em = a.elen, ew = (em < 64 && (em & (em - 1)) == 0) ? em : 64;
        int e_opaque0 = 0;                               // keeps the term offsets from being hoisted out of the filter loop
        asm volatile("s_mov_b32 %0, 0" : "=s"(e_opaque0));
        const int e_lw = 31 - __builtin_clz(ew);         // ew is a power of two: shifts, not divisions
        const int e_lane = (tid & 63) + e_opaque0, e_sub = e_lane >> e_lw, e_li = e_lane - (e_sub << e_lw);
        const double* ewt_f = a.ewt + a.ewt_off[f];
        double w_last0 = 0.0;
        if (band_energy && e_li < em) w_last0 = ewt_f[e_li + ((((em - 1 - e_li) >> e_lw) < 15 ? ((em - 1 - e_li) >> e_lw) : 15) << e_lw)];
        // Y = X H packed for the inverse: with A = Y[k], B = conj Y[M-k], S = A + B, t = conj(w^k) (A - B) the inputs are
        // conj((S + i t) / 2) at k and conj((conj S + i conj t) / 2) at M - k; they go to their owners through LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = tid + j * kObThreads;
            C A = cmul(xa[j], H[(FRT_OB_ABLATE & 2) ? j : k]);
            C Bm = cmul(xb[j], H[(FRT_OB_ABLATE & 2) ? j + 4 : M - k]);
            if (k == 0) { A.y = 0.0; Bm.y = 0.0; }              // irfft ignores the imaginary part of the edge bins
            const C B = cconj(Bm);
            const C S = A + B, D = A - B;
            const C t = cmul(cconj(twl[k]), D);
            other[k] = {0.5 * (S.x - t.y), -0.5 * (S.y + t.x)};
            if (k != 0) other[M - k] = {0.5 * (S.x + t.y), 0.5 * (S.y - t.x)};
        }
        if (tid == 0) {
            const C A = cmul(xmid_lds, H[M / 2]);                    // k = M/2: B = conj A, conj(w^k) = i: t = i (0, 2 A.y)
            other[M / 2] = {A.x, A.y};                             // conj((S + i t)/2) = conj((A.x, -A.y)) ... = (A.x, A.y)
        }
        __syncthreads();                                         // xmid_lds (first filter) and the packed inputs are visible
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = other[tid + j * kObThreads];
        // exchanges in buf, other, buf: every thread is past the barrier above, so nobody reads `buf` (the previous filter's
        // window, or Z) any more; conj(FFT(conj Z)) = M ifft(Z)
        fft2048_two_buffers(v, buf, other, tid, twt);
        double* out = (double*)other;                            // the finished window, plain doubles (its readers: the gathers
                                                                 // of the second exchange, two barriers back)
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
        } else if (!(FRT_OB_ABLATE & 4)) {
            if (a.y) {
                double* y = a.y + (long long)c * a.y_cstride + a.y_off[f] + o0;
                for (int t = tid; t < Lb; t += kObThreads) y[t] = res[t];
            }
            if (a.eblock) {
                // zero-state block energies alpha sum_i (1-alpha)^(m-1-i) y_i^2 (exp_smoothing.py:40-56): groups of
                // w = min(m, 64) lanes per energy block, fixed summation order
                // (a power of two below 64: that many lanes per block and several blocks per wave; anything else: a whole wave)
                const int m = em, w = ew, per_wave = 64 / w;
                const int wave = tid >> 6, sub = e_sub, li = e_li;
                const double rinv = a.ewr[f];
                const int ne = Lb / m;
                double* eo = a.eblock + ((size_t)c * a.nblocks + (size_t)(o0 / m)) * a.nbands + a.band_index[f];
                for (int b0 = 0; b0 < m; b0 += 16 * w) {                     // one trip unless m > 1024 (whole-chunk blocks)
                    // the lane's weights of this trip: terms u = 0..umax, from the last one down
                    const int umax = li + b0 < m ? (((m - 1 - li - b0) >> e_lw) < 15 ? ((m - 1 - li - b0) >> e_lw) : 15) : -1;
                    double wv[16];
                    double wcur = b0 == 0 ? w_last0 : (umax >= 0 ? ewt_f[b0 + li + (umax << e_lw)] : 0.0);
#pragma unroll
                    for (int u = 15; u >= 0; --u) {
                        wv[u] = u > umax ? 0.0 : wcur;
                        if (u <= umax && !(FRT_OB_EABL & 8)) wcur *= rinv;
                    }
                    for (int e0 = wave * per_wave; e0 < ne; e0 += (kObThreads / 64) * per_wave) {
                        const int le = e0 + sub;
                        // branch-free: a term past the block's end has weight 0 and reads a clamped (valid) address
                        double acc = 0.0;
#pragma unroll
                        for (int h = 0; h < 16; h += 8) {                    // eight reads in flight
                            double val[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int at = le * m + b0 + li + ((h + u) << e_lw);
                                val[u] = (FRT_OB_EABL & 4) ? (double)at : res[at < F - kTail ? at : F - kTail - 1];
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) acc += wv[h + u] * (val[u] * val[u]);
                        }
                        if (!(FRT_OB_EABL & 2)) acc = group_sum(acc, w);
                        if (le < ne && li == 0 && (!(FRT_OB_EABL & 1) || acc == -1.0)) {
                            if (b0 == 0) eo[(size_t)le * a.nbands] = acc;
                            else eo[(size_t)le * a.nbands] += acc;           // (m > 1024: same thread, same order every run)
                        }
                    }
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
        // no barrier: the next filter packs into `buf` (its last readers, the third exchange's gathers, are behind the barrier
        // above), and the array holding this window is written again only behind the next filter's first barrier
        C* const tmp = buf;
        buf = other;
        other = tmp;
    }
}

#endif  // FRT_EXPERIMENTS

#include "ola_wave.h"
#if FRT_OW_TIMING
extern "C" int frt_ow_timing_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(frt::ow_timing), sizeof(frt::ow_timing)) == hipSuccess ? 0 : -1;
}
#endif

static int ola_batch_tables(frt_octbank* h) {
    frt_ola_state* o = h->ola;
    if (o->bHw.ptr) return FRT_OK;
    int rc;
    // H_f[k] = sum_t h_f[t] exp(-2 pi i k t / F): the rfft of the zero-padded taps (filter_design.py computes the same
    // with numpy at the reference's own sizes)
    const int nfilt = h->nfilt;
    std::vector<long double> ct(kObF), st(kObF);
    const long double pi2 = 6.283185307179586476925286766559L;
    for (int t = 0; t < kObF; ++t) {
        ct[t] = cosl(pi2 * t / kObF);
        st[t] = sinl(pi2 * t / kObF);
    }
    if ((rc = upload(o->btw, make_twiddles<double>(kObM)))) return rc;          // exp(-2 pi i t / 2048): both kernels' transforms
#ifdef FRT_EXPERIMENTS
    if ((rc = upload(o->btwl, make_twiddles<double>(kObF, kObM + 1)))) return rc;
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
#endif
    // ola_pair_kernel: every bin of the 2048-point transform (H[k] = H4096[2 k]), conjugated and scaled for the inverse
    std::vector<double> Hw((size_t)nfilt * kOwN * 2);
    for (int f = 0; f < nfilt; ++f) {
        const double* taps = &o->h_taps[(size_t)f * kFirLength];
        for (int k = 0; k < kOwN; ++k) {
            long double re = 0, im = 0;
            for (int t = 0; t < kFirLength; ++t) {
                const int idx = (2 * k * t) & (kObF - 1);
                re += taps[t] * ct[idx];
                im -= taps[t] * st[idx];
            }
            Hw[((size_t)f * kOwN + k) * 2] = (double)(re / kOwN);
            Hw[((size_t)f * kOwN + k) * 2 + 1] = (double)(-im / kOwN);
        }
    }
    if ((rc = upload(o->bHw, Hw))) return rc;
    if ((rc = o->pending_next.reserve(o->pending.bytes))) return rc;
    return FRT_OK;
}

// smoothing weights per band: alpha (1 - alpha)^(m - 1 - i), m = the band's samples per energy block — eblock0 / dec, or
// (whole) the band's stage length of an n-sample chunk (exp_smoothing.py:40-56 with the kernels of octavespectrum.py:77-81)
static int ola_energy_weights(frt_octbank* h, int64_t n, int eblock0, bool whole, const double* alphas) {
    frt_ola_state* o = h->ola;
    bool same = o->ewt_block == eblock0 && (int)o->ewt_alpha.size() == h->nbands && (!whole || o->ewt_n == (int)n);
    for (int k = 0; same && k < h->nbands; ++k) same = o->ewt_alpha[k] == alphas[k];
    if (same) return FRT_OK;
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
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));      // the old tables may still be read
    int rc;
    if ((rc = upload(o->ewt, wt)) || (rc = upload(o->ewt_off_dev, o->ewt_off))) return rc;
    o->ewt_block = eblock0;
    o->ewt_n = whole ? (int)n : -1;
    o->ewt_alpha.assign(alphas, alphas + h->nbands);
    return FRT_OK;
}

int frt_ola_filter_batch(frt_octbank* h, const void* d_x, int x_f32, int64_t n, double* d_y, int64_t y_cstride,
                         double* d_eblock, int eblock0, int nblocks, const double* alphas) {
    frt_ola_state* o = h->ola;
    // one energy block = the whole call (the widget's chunk, any length): a band's block is its stage's whole output
    const bool whole = d_eblock && nblocks == 1 && eblock0 == n;
    // (served by the chunk kernels below; -DFRT_EXPERIMENTS builds keep round 2's kernel for it and for FRT_OLA_NO_WAVE)
#ifdef FRT_EXPERIMENTS
    const bool use_wave = !whole && exp_env("FRT_OLA_NO_WAVE") == nullptr;
#else
    FRT_REQUIRE(!whole, "frt_ola_filter_batch: a call of one block belongs to the chunk kernels (frt_ola_chunk_energies)");
    constexpr bool use_wave = true;
#endif
    int rc;
    if ((rc = ola_batch_tables(h))) return rc;
    long long len[kNOctave];
    len[0] = n;
    for (int j = 1; j < kNOctave; ++j) len[j] = (len[j - 1] + 1) / 2;
    for (int j = 1; j < kNOctave; ++j)
        if ((rc = h->xbuf[j].reserve((size_t)h->n_channels * len[j] * sizeof(double)))) return rc;
    std::vector<long long> band_off(h->nbands + 1, 0);
    for (int k = 0; k < h->nbands; ++k) band_off[k + 1] = band_off[k] + len[kNOctave - 1 - k / h->bpo];
    if (d_eblock && (rc = ola_energy_weights(h, n, eblock0, whole, alphas))) return rc;
    const size_t stage_pend = (size_t)h->n_channels * h->nfilt * kTail;
    std::vector<OlaBatchArgs> deferred;            // band filters of the stages whose decimators ran ahead (see below)
    std::vector<long long> deferred_sets;
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
            if (d_eblock) {
                const int ew = (a.elen < 64 && (a.elen & (a.elen - 1)) == 0) ? a.elen : 64;      // lanes per energy block
                a.ewr[i] = std::pow(1.0 - o->ewt_alpha[band], (double)ew);
            }
        }
        a.xnext = j + 1 < kNOctave ? h->xbuf[j + 1].as<double>() : nullptr;
        a.xnext_stride = j + 1 < kNOctave ? len[j + 1] : 0;
        if (use_wave) {
            // one workgroup of ola_pair_kernel per set of 3072 outputs; the stage has len + 511 of them (the last 511 are the new
            // tails).  Filter groups repeat the forward transform: as many as make the launch's rounds x transforms per
            // workgroup smallest (four workgroups fit a CU)
            a.Hw = o->bHw.as<double>();
            for (int i = 0; i < h->bpo; ++i) a.er[i] = d_eblock ? 1.0 - o->ewt_alpha[a.band_index[i]] : 0.0;
            const long long nsets = (len[j] + kTail + kOwSet - 1) / kOwSet, slots = 4ll * device_cu_count();
            auto launch = [&](int f_first, int f_count) {
                long long best = -1;
                int best_groups = 1;
                for (int g = 1; g <= f_count; ++g) {
                    const int gs = (f_count + g - 1) / g, g2 = (f_count + gs - 1) / gs;
                    const long long rounds = (nsets * h->n_channels * g2 + slots - 1) / slots, cost = rounds * (1 + gs);
                    if (best < 0 || cost < best) {
                        best = cost;
                        best_groups = g2;
                    }
                }
                a.f_first = f_first;
                a.f_count = f_count;
                a.gsize = (f_count + best_groups - 1) / best_groups;
                hipLaunchKernelGGL(ola_pair_kernel, dim3((unsigned)nsets, best_groups, h->n_channels), dim3(kOwThreads), 0, h->stream, a);
            };
            // Only the DECIMATOR of a stage feeds the next one.  With many band filters per stage (bpo >= 6) the decimators run
            // ahead, one short launch per stage (1 + 1 transforms per set instead of 1 + nfilt), and the band filters of those
            // stages — which nothing waits for — follow in ONE launch over all of them (`deferred`, ola_pair_multi_kernel): the
            // low-rate stages stop being a chain of launches that no longer fill the chip (15-50 us each in round 4), at the price
            // of one more forward transform per set.  Measured (profiles/r05_ola_defer.txt, sets x channels up to 4 rounds of the
            // chip's workgroup slots deferred): 8 ch x 2^20 at bpo 24 1.267 -> 1.137 ms, bpo 12 0.767 -> 0.688, bpo 6 0.768 -> 0.727;
            // with 4 filters per stage (bpo 3) the repeated forward transform costs what the shorter chain saves (0.789 -> 0.799)
            // and with 2 (bpo 1) more: one launch per stage there.
            long long defer_below = h->bpo >= 6 ? 4 * slots : 0;
            if (const char* e = exp_env("FRT_OLA_DEFER_BELOW")) defer_below = atoll(e);
            // (not while the caller's stream is being captured: the deferred launch's argument table is uploaded by a copy whose node
            // would keep pointing at this handle's staging block; option "ola_defer" = 0: tests, a launch per stage)
            const bool split_stage = nsets * h->n_channels <= defer_below && exp_env("FRT_OLA_NO_DEFER") == nullptr &&
                                     option(kOptOlaDefer) != 0 && !CaptureScope::active();
            if (!split_stage) {
                launch(0, h->nfilt);
            } else {
                launch(h->bpo, 1);
                deferred.push_back(a);
                deferred_sets.push_back(nsets);
            }
            FRT_HIP_CHECK(hipGetLastError());
            continue;
        }
#ifdef FRT_EXPERIMENTS
        const long long nblk = (len[j] + kObL - 1) / kObL;
        // filter groups: every workgroup repeats the forward transform of its window, so as few groups as still fill the chip
        int groups = 1;
        while (groups < h->nfilt && nblk * h->n_channels * groups < 2ll * device_cu_count()) ++groups;
        a.gsize = (h->nfilt + groups - 1) / groups;
        groups = (h->nfilt + a.gsize - 1) / a.gsize;
        hipLaunchKernelGGL(ola_batch_kernel, dim3((unsigned)nblk, groups, h->n_channels), dim3(kObThreads), 0, h->stream, a);
        FRT_HIP_CHECK(hipGetLastError());
#endif
    }
    if (!deferred.empty()) {
        // the deferred band filters: ONE launch over all their stages (ola_pair_multi_kernel).  Its argument table lives in device
        // memory; it changes only with the call's buffers and with the parity of the tails' swap, so a steady stream of calls
        // uploads it a few times (a byte-wise mismatch — padding included — costs an upload, never a wrong table).  The upload
        // is staged through a page-locked block of the handle: an asynchronous copy reads its source when it executes.
        OlaMultiIndex m{};
        m.nstage = (int)deferred.size();
        int best_groups = 1;
        {
            long long total = 0, best = -1;
            for (long long ns : deferred_sets) total += ns;
            const long long slots = 4ll * device_cu_count();
            for (int g = 1; g <= h->bpo; ++g) {
                const int gs = (h->bpo + g - 1) / g, g2 = (h->bpo + gs - 1) / gs;
                const long long rounds = (total * h->n_channels * g2 + slots - 1) / slots, cost = rounds * (1 + gs);
                if (best < 0 || cost < best) {
                    best = cost;
                    best_groups = g2;
                }
            }
        }
        int at = 0;
        for (size_t i = 0; i < deferred.size(); ++i) {
            m.first_set[i] = at;
            at += (int)deferred_sets[i];
            deferred[i].f_first = 0;
            deferred[i].f_count = h->bpo;
            deferred[i].gsize = (h->bpo + best_groups - 1) / best_groups;
        }
        m.first_set[deferred.size()] = at;
        const int parity = o->multi_parity;
        const size_t bytes = deferred.size() * sizeof(OlaBatchArgs);
        std::vector<char>& held = o->multi_host[parity];
        if (held.size() != bytes || memcmp(held.data(), deferred.data(), bytes) != 0) {
            if ((rc = o->multi_tab[parity].reserve(kNOctave * sizeof(OlaBatchArgs)))) return rc;
            if (!o->multi_pin[parity]) FRT_HIP_CHECK(hipHostMalloc(&o->multi_pin[parity], kNOctave * sizeof(OlaBatchArgs), hipHostMallocDefault));
            else FRT_HIP_CHECK(hipStreamSynchronize(h->stream));      // (an earlier upload from this block may still be queued)
            held.assign((const char*)deferred.data(), (const char*)deferred.data() + bytes);
            memcpy(o->multi_pin[parity], held.data(), bytes);
            FRT_HIP_CHECK(hipMemcpyAsync(o->multi_tab[parity].ptr, o->multi_pin[parity], bytes, hipMemcpyHostToDevice, h->stream));
        }
        hipLaunchKernelGGL(ola_pair_multi_kernel, dim3((unsigned)at, best_groups, h->n_channels), dim3(kOwThreads), 0, h->stream,
                           o->multi_tab[parity].as<OlaBatchArgs>(), m);
        FRT_HIP_CHECK(hipGetLastError());
        o->multi_parity ^= 1;
    }
    std::swap(o->pending.ptr, o->pending_next.ptr);             // equal sizes; the streaming path and the graphs follow `pending`
    return FRT_OK;
}


// ---- chunk path: ONE block of 1..1024 samples, the octave-spectrum widget's handler (octavespectrum.py:91-122) ------------
// The streaming object pushes one audio chunk (512 samples) at a time and wants 9 x bpo numbers back.  Through the
// transform kernels above that is nine DEPENDENT launches (the stages chain through the decimated signal) of ~10 us each,
// whatever the stage length — a 4096-point window to filter the two samples of stage 8.  At these sizes the running
// convolution itself is cheaper than its FFT form: a stage's block contributes
//     contrib[t] = sum_k h[k] x[t - k],   0 <= t < m + 511
// to its own outputs (t < m, plus the carried tail pending[t]) and to the next tail (pending'[t'] = contrib[m + t'] +
// pending[m + t']) — the same sums the overlap-add forms through X H (filter.py:213-245), in float64, to rounding.  Two launches:
//   A  one workgroup per channel walks the DECIMATOR through stages 0..7: only the even outputs (the next stage's input,
//      y_dec[:m:2]) and the new tail, 0.43 M multiply-adds for a 512-sample chunk;
//   B  one workgroup per (stage, band filter, channel): outputs, tail, the block's smoothed energy
//      sum_t alpha (1 - alpha)^(m-1-t) y[t]^2 + (1 - alpha)^m sp_prev (exp_smoothing.py:40-56) and the dB value.
// A thread owns a PAIR of consecutive outputs (t even): per two taps it reads one aligned pair of input samples from LDS
// (the previous pair stays in registers) for four multiply-adds; the taps are wave-uniform and come through the scalar
// cache.  The stage input sits in LDS between zeros, so no tap needs a range test; a wavefront only walks the taps that
// can meet a sample for one of its outputs.
constexpr int kOcThreads = 1024;
constexpr int kOcFront = kFirLength - 1;                 // zeros in front of the samples: t - k >= -511
constexpr int kOcMaxN = 1024;
constexpr int kOcPadLen = kOcFront + kOcMaxN + kFirLength + 1;      // ... and behind: t + 2 - k <= m + 511
static_assert(kOcPadLen % 2 == 0 && kOcFront % 2 == 1, "pairs (x[t-k-1], x[t-k]) start at even LDS indices for even t, k");

struct OlaChunkArgs {
    const void* x;             // [C][x_stride] chunk: float (x_f32) or double; device-accessible (HBM or pinned host)
    int x_f32;
    long long x_stride;
    int len[kNOctave];         // stage lengths
    int xoff[kNOctave];        // stage inputs inside a channel's row of xs
    double* xs;                // [C][xs_stride]
    long long xs_stride;
    const double* taps;        // [nfilt][512]
    double* pending;           // [9][C][nfilt][511]
    int nfilt, bpo, n_channels, nbands;
    const double* ewt;         // alpha (1 - alpha)^(m - 1 - t) per band at ewt_off
    const long long* ewt_off;
    const double* decay_n;     // (1 - alpha)^m per band
    double* smooth;            // [C][nbands] carried smoothed energies
    const double* weight_db;   // [nbands] or null
    int as_db;
    void* out;                 // [C][nbands]
    int out_f32;
    double* y;                 // band signals, packed per channel (band k after band k - 1), or null
    long long y_cstride;
};

typedef const double __attribute__((address_space(4))) * oc_ktable;

// (y0, y1) = contrib[t], contrib[t + 1] for even t over the taps k0 .. k1 + 7 (k0, k1 multiples of 8, wave-uniform).
// Eight taps per trip: four aligned sample pairs and the eight taps are fetched for the NEXT trip while this trip's sixteen
// multiply-adds issue (one pair and two taps per trip left every load's latency in the open).
template <bool BOTH>
__device__ __forceinline__ void oc_fir_pair(const double* xpad, int t, oc_ktable h, int k0, int k1, double& y0, double& y1) {
    const double2* p = (const double2*)(xpad + (kOcFront + t - k0 - 1));       // (x[t-k-1], x[t-k]) at k = k0
    double2 hi = p[1];                                                           // (x[t-k+1], x[t-k+2])
    double2 lo[4];
    double hk[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) lo[q] = p[-q];
#pragma unroll
    for (int q = 0; q < 8; ++q) hk[q] = h[k0 + q];
    double a0 = 0.0, a1 = 0.0;
    for (int k = k0; k <= k1; k += 8) {
        double2 cur[4];
        double hc[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = lo[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) hc[q] = hk[q];
        p -= 4;
        if (k + 8 <= k1) {                                   // uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) lo[q] = p[-q];
#pragma unroll
            for (int q = 0; q < 8; ++q) hk[q] = h[k + 8 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0 = __builtin_fma(hc[2 * q], cur[q].y, a0);
            a0 = __builtin_fma(hc[2 * q + 1], cur[q].x, a0);
            if (BOTH) {
                a1 = __builtin_fma(hc[2 * q], hi.x, a1);
                a1 = __builtin_fma(hc[2 * q + 1], cur[q].y, a1);
            }
            hi = cur[q];
        }
    }
    y0 = a0;
    y1 = a1;
}

// tap range of a wavefront whose outputs span [t_lo, t_hi + 1] (even t_lo, t_hi): tap k meets a sample iff 0 <= t - k < m;
// widened to whole groups of eight taps (the samples the extra taps meet are the zeros around the stage input)
__device__ __forceinline__ void oc_tap_range(int t_lo, int t_hi, int m, int& k0, int& k1) {
    k0 = t_lo + 1 - m;
    k0 = k0 < 0 ? 0 : (k0 & ~7);
    k1 = t_hi + 1 < kFirLength - 1 ? t_hi + 1 : kFirLength - 1;      // last tap that meets a sample
    k1 &= ~7;                                                      // first tap of the last group
    k0 = __builtin_amdgcn_readfirstlane(k0);
    k1 = __builtin_amdgcn_readfirstlane(k1);
}

__device__ __forceinline__ void oc_stage_input(double* xpad, double* pend, const double* x, int m, const double* pend_g, int tid) {
    for (int i = tid; i < kOcPadLen; i += kOcThreads) {
        const int s = i - kOcFront;
        xpad[i] = (s >= 0 && s < m) ? x[s] : 0.0;
    }
    for (int i = tid; i < kFirLength; i += kOcThreads) pend[i] = i < kTail ? pend_g[i] : 0.0;
}

// Launch A: the decimator's even outputs, stage after stage, nothing else — its new tails do not feed the chain and are
// left to launch B.  The chain is a critical path (stage j + 1 waits for stage j), so a stage's outputs are spread over all
// sixteen wavefronts: groups of 64 outputs x S parts of the group's tap range (a wavefront's taps stay uniform), the parts
// summed in a fixed order afterwards.  The carried tails of all eight stages are fetched once, up front; the next stage's
// input is written straight back into the LDS array (and to xs for launch B, not waited for).
__global__ void __launch_bounds__(kOcThreads) ola_chunk_dec_kernel(const OlaChunkArgs a) {
    __shared__ __attribute__((aligned(16))) double xpad[kOcPadLen];
    __shared__ double pend[kNOctave - 1][kFirLength];
    __shared__ double part[kOcThreads];
    const int tid = threadIdx.x, c = blockIdx.x, wave = tid >> 6, lane = tid & 63;
    const int dec = a.bpo;
    const oc_ktable h = (oc_ktable)(uintptr_t)(a.taps + (size_t)dec * kFirLength);
    double* xs = a.xs + (size_t)c * a.xs_stride;
    {
        const int m = a.len[0];
        for (int i = tid; i < kOcPadLen; i += kOcThreads) {
            const int s = i - kOcFront;
            double v = 0.0;
            if (s >= 0 && s < m) {
                const long long g = (long long)c * a.x_stride + s;
                v = a.x_f32 ? (double)((const float*)a.x)[g] : ((const double*)a.x)[g];
                xs[a.xoff[0] + s] = v;                    // the chunk as float64 for launch B
            }
            xpad[i] = v;
        }
        for (int i = tid; i < (kNOctave - 1) * kFirLength; i += kOcThreads) {
            const int j = i / kFirLength, t = i - j * kFirLength;
            pend[j][t] = t < kTail ? a.pending[(((size_t)j * a.n_channels + c) * a.nfilt + dec) * kTail + t] : 0.0;
        }
    }
    __syncthreads();
    for (int j = 0; j + 1 < kNOctave; ++j) {
        const int m = a.len[j], nh = (m + 1) / 2;
        const int G = (nh + 63) / 64;                        // <= 8
        const int S = G > 4 ? 2 : G > 2 ? 4 : G > 1 ? 8 : 16;
        const int g = wave / S, sp = wave - g * S;
        double y0 = 0.0, y1;
        if (g < G) {
            const int u0 = g * 64, u = u0 + lane;
            const int ul = (u0 + 63 < nh ? u0 + 63 : nh - 1);
            int k0, k1;
            oc_tap_range(2 * u0, 2 * ul, m, k0, k1);
            const int trips = (k1 - k0) / 8 + 1, per = (trips + S - 1) / S;
            const int ka = k0 + 8 * per * sp;
            int kb = ka + 8 * (per - 1);
            kb = kb < k1 ? kb : k1;
            if (ka <= kb) oc_fir_pair<false>(xpad, 2 * (u < nh ? u : nh - 1), h, ka, kb, y0, y1);
        }
        part[tid] = y0;
        __syncthreads();                                     // every read of xpad is done
        {
            // thread u sums its output's parts; everybody clears what the shorter next stage no longer covers
            const int u = tid;
            if (u < nh) {
                const int gg = u >> 6, l = u & 63, t = 2 * u;
                double v = 0.0;
                for (int q = 0; q < S; ++q) v += part[(gg * S + q) * 64 + l];
                v += pend[j][t < kTail ? t : kTail];
                xs[a.xoff[j + 1] + u] = v;
                xpad[kOcFront + u] = v;
            }
            for (int i = nh + tid; i < m; i += kOcThreads) xpad[kOcFront + i] = 0.0;
        }
        __syncthreads();
    }
}

// Launch B: one workgroup per (stage, band filter, channel) — outputs, new tail, smoothed energy, dB — plus eight per channel
// for the decimator's new tails (stages 0..7).
__global__ void __launch_bounds__(kOcThreads) ola_chunk_band_kernel(const OlaChunkArgs a) {
    __shared__ __attribute__((aligned(16))) double xpad[kOcPadLen];
    __shared__ double pend[kFirLength];
    __shared__ double red[kOcThreads / 64];
    const int tid = threadIdx.x, c = blockIdx.y, wave = tid >> 6, lane = tid & 63;
    const int nb = kNOctave * a.bpo;
    const bool is_dec = (int)blockIdx.x >= nb;
    const int j = is_dec ? (int)blockIdx.x - nb : (int)blockIdx.x / a.bpo;
    const int f = is_dec ? a.bpo : (int)blockIdx.x - j * a.bpo;
    const int m = a.len[j];
    const int band = is_dec ? 0 : (kNOctave - 1 - j) * a.bpo + f;
    const oc_ktable h = (oc_ktable)(uintptr_t)(a.taps + (size_t)f * kFirLength);
    double* pend_g = a.pending + (((size_t)j * a.n_channels + c) * a.nfilt + f) * kTail;
    oc_stage_input(xpad, pend, a.xs + (size_t)c * a.xs_stride + a.xoff[j], m, pend_g, tid);
    __syncthreads();
    const int t_first = is_dec ? (m & ~1) : 0;             // the decimator's outputs below m were launch A's
    const int np = (m + kTail - t_first + 1) / 2;          // pairs covering [t_first, m + 511)
    double e = 0.0;
    if (wave < (np + 63) / 64) {
        const int v0 = wave * 64, v = v0 + lane;
        const int vl = (v0 + 63 < np ? v0 + 63 : np - 1);
        int k0, k1;
        oc_tap_range(t_first + 2 * v0, t_first + 2 * vl, m, k0, k1);
        const int t = t_first + 2 * (v < np ? v : np - 1);
        double y[2];
        oc_fir_pair<true>(xpad, t, h, k0, k1, y[0], y[1]);
        if (v < np) {
            const bool energy = !is_dec && a.ewt != nullptr;
            const double* w = energy ? a.ewt + a.ewt_off[band] : nullptr;
            // a channel's packed row holds band k after band k - 1, bands of the lowest-rate stage first (filter.py:239-245)
            double* yrow = nullptr;
            if (a.y && !is_dec) {
                long long off = (long long)f * m;
                for (int jj = j + 1; jj < kNOctave; ++jj) off += (long long)a.bpo * a.len[jj];
                yrow = a.y + (long long)c * a.y_cstride + off;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int tt = t + r;
                const double val = y[r] + pend[tt < kTail ? tt : kTail];
                if (tt < m) {
                    if (energy) e = __builtin_fma(w[tt], val * val, e);
                    if (yrow) yrow[tt] = val;
                } else if (tt < m + kTail) pend_g[tt - m] = val;
            }
        }
    }
    if (is_dec || a.ewt == nullptr) return;                // uniform
    // the block's energy: lanes, then wavefronts, in a fixed order
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o, 64);
    if (lane == 0) red[wave] = e;
    __syncthreads();
    if (tid == 0) {
        double E = 0.0;
        for (int w = 0; w < kOcThreads / 64; ++w) E += red[w];
        double* sm = a.smooth + (size_t)c * a.nbands + band;
        const double sp = E + *sm * a.decay_n[band];                 // exp_smoothing.py:52-54
        *sm = sp;
        double v = sp;
        if (a.as_db) v = 10.0 * log10(sp + 1e-30) + (a.weight_db ? a.weight_db[band] : 0.0);
        const size_t o = (size_t)c * a.nbands + band;
        if (a.out_f32) ((float*)a.out)[o] = (float)v;
        else ((double*)a.out)[o] = v;
    }
}

int frt_ola_chunk_energies(frt_octbank* h, const void* x, int x_f32, int n, const double* alphas, const double* d_decay_n,
                           double* d_smooth, const double* d_weight_db, int as_db, void* out, int out_f32) {
    frt_ola_state* o = h->ola;
    FRT_REQUIRE(n >= 1 && n <= kOcMaxN, "frt_ola_chunk_energies: n %d not in [1, %d]", n, kOcMaxN);
    int rc;
    if (!o->taps.ptr && (rc = upload(o->taps, o->h_taps))) return rc;
    if ((rc = ola_energy_weights(h, n, n, true, alphas))) return rc;
    OlaChunkArgs a{};
    a.x = x;
    a.x_f32 = x_f32;
    a.x_stride = n;
    int total = 0;
    for (int j = 0, m = n; j < kNOctave; ++j, m = (m + 1) / 2) {
        a.len[j] = m;
        a.xoff[j] = total;
        total += (m + 1) & ~1;
    }
    a.xs_stride = total;
    if ((rc = o->xs.reserve((size_t)h->n_channels * total * sizeof(double)))) return rc;
    a.xs = o->xs.as<double>();
    a.taps = o->taps.as<double>();
    a.pending = o->pending.as<double>();
    a.nfilt = h->nfilt;
    a.bpo = h->bpo;
    a.n_channels = h->n_channels;
    a.nbands = h->nbands;
    a.ewt = o->ewt.as<double>();
    a.ewt_off = o->ewt_off_dev.as<long long>();
    a.decay_n = d_decay_n;
    a.smooth = d_smooth;
    a.weight_db = d_weight_db;
    a.as_db = as_db;
    a.out = out;
    a.out_f32 = out_f32;
    hipLaunchKernelGGL(ola_chunk_dec_kernel, dim3(h->n_channels), dim3(kOcThreads), 0, h->stream, a);
    hipLaunchKernelGGL(ola_chunk_band_kernel, dim3(kNOctave * h->bpo + kNOctave - 1, h->n_channels), dim3(kOcThreads), 0, h->stream, a);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

// The same two launches for a call that wants the band SIGNALS of one block of 1..1024 samples (Octave_Filters.filter,
// octavefilters.py:49-58): x [C][n] float64 and y packed [C][y_cstride], both device accessible.
int frt_ola_chunk_filter(frt_octbank* h, const double* x, int n, double* y, int64_t y_cstride) {
    frt_ola_state* o = h->ola;
    FRT_REQUIRE(n >= 1 && n <= kOcMaxN, "frt_ola_chunk_filter: n %d not in [1, %d]", n, kOcMaxN);
    int rc;
    if (!o->taps.ptr && (rc = upload(o->taps, o->h_taps))) return rc;
    OlaChunkArgs a{};
    a.x = x;
    a.x_f32 = 0;
    a.x_stride = n;
    int total = 0;
    for (int j = 0, m = n; j < kNOctave; ++j, m = (m + 1) / 2) {
        a.len[j] = m;
        a.xoff[j] = total;
        total += (m + 1) & ~1;
    }
    a.xs_stride = total;
    if ((rc = o->xs.reserve((size_t)h->n_channels * total * sizeof(double)))) return rc;
    a.xs = o->xs.as<double>();
    a.taps = o->taps.as<double>();
    a.pending = o->pending.as<double>();
    a.nfilt = h->nfilt;
    a.bpo = h->bpo;
    a.n_channels = h->n_channels;
    a.nbands = h->nbands;
    a.y = y;
    a.y_cstride = y_cstride;
    hipLaunchKernelGGL(ola_chunk_dec_kernel, dim3(h->n_channels), dim3(kOcThreads), 0, h->stream, a);
    hipLaunchKernelGGL(ola_chunk_band_kernel, dim3(kNOctave * h->bpo + kNOctave - 1, h->n_channels), dim3(kOcThreads), 0, h->stream, a);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}
