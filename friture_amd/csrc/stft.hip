// stft.hip — K1: batched short-time Fourier transform -> power spectrum for gfx950.
//
// Reference semantics (all citations relative to the reference checkout):
//   P[k] = |rfft(x * w)[k]|^2 / N^2, w = symmetric Hann        friture/audioproc.py:42-50,76-80
//   frame loop with hop = int(N * (1 - overlap))                 friture/spectrum.py:144-155,
//                                                                friture/spectrogram.py:149-159
//   dB = 10 log10(P + 1e-30) + weight[k]                         friture/spectrogram.py:119-125
//   norm = (dB - spec_min) / (spec_max - spec_min)               friture/spectrogram.py:128-129
//   pixel = lut[int(clip(norm, 0, 1) * 255)]                     friture/signal/color_tranform.py:48-51,
//                                                                friture/signal/lookup_table.py:50-52
//
// Kernel shape.  A frame of N real samples is an M = N/2 point complex FFT served by TPF = M/8
// threads holding 8 points each (fft_core.h).  A "lane group" of TPF threads walks a run of
// consecutive frames of one channel.  With hop = s*N/8 the last 8-s register slots of a frame are
// the first 8-s slots of the next one, so each input sample is fetched from HBM once per run and
// the only HBM traffic is 4*hop bytes in and 4*(N/2+1) bytes out per spectrum.  The samples of the
// next frame are requested before the current frame is transformed (one frame of prefetch per
// wave).  For N <= 1024 a frame lives inside one wavefront: pass exchanges need no barrier and the
// conjugate-symmetric unpack is done with wavefront shuffles instead of LDS.
#include "stft_wave.h"

#include "stft_big.h"
#include "stft_pk.h"          // packed-arithmetic helpers; round 3's N = 16384 instance itself only in -DFRT_EXPERIMENTS builds
#include "stft_pk16.h"
#ifdef FRT_EXPERIMENTS
#include "../../tools/exp/stft_pk16r.h"      // N = 16384 with two workgroups per CU: measured 7-25 % slower (profiles/r05_stft16384_two_workgroups.txt)
#endif
#include "stft_pk16s.h"       // N = 8192 / 4096 / 2048: one template over the size

namespace frt {

#ifdef FRT_EXPERIMENTS
// N = 16384, hop N/2 or N/4, rows on 16-byte boundaries: the packed-arithmetic instance (stft_pk.h)
template <int HS>
static int launch_pk(const StftArgs& a, hipStream_t stream) {
    const dim3 grid(a.n_groups), block(PkPlan::BLOCK);
    switch (a.kind) {
        case FRT_STFT_PSD: hipLaunchKernelGGL((stft_pk_kernel<0, HS>), grid, block, 0, stream, a); break;
        case FRT_STFT_IMAGE:
            if (a.eps_free) hipLaunchKernelGGL((stft_pk_kernel<4, HS>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((stft_pk_kernel<3, HS>), grid, block, 0, stream, a);
            break;
        default: hipLaunchKernelGGL((stft_pk_kernel<1, HS>), grid, block, 0, stream, a); break;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}
#endif

// the same with the sub-transforms factored 16 x 16 x 2 (stft_pk16.h)
template <int HS>
static int launch_pk16(const StftArgs& a, hipStream_t stream) {
    const dim3 grid(a.n_groups), block(Pk16Plan::BLOCK);
    switch (a.kind) {
        case FRT_STFT_PSD: hipLaunchKernelGGL((stft_pk16_kernel<0, HS>), grid, block, 0, stream, a); break;
        case FRT_STFT_IMAGE:
            if (a.eps_free) hipLaunchKernelGGL((stft_pk16_kernel<4, HS>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((stft_pk16_kernel<3, HS>), grid, block, 0, stream, a);
            break;
        default: hipLaunchKernelGGL((stft_pk16_kernel<1, HS>), grid, block, 0, stream, a); break;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

#ifdef FRT_EXPERIMENTS
// N = 16384 with two workgroups per CU: samples in a register ring, constants streamed (stft_pk16r.h; FRT_STFT_PK16R=1 selects it)
template <int HS>
static int launch_pk16r(const StftArgs& a, hipStream_t stream) {
    const dim3 grid(a.n_groups), block(Pk16rPlan::BLOCK);
    switch (a.kind) {
        case FRT_STFT_PSD: hipLaunchKernelGGL((stft_pk16r_kernel<0, HS>), grid, block, 0, stream, a); break;
        case FRT_STFT_IMAGE:
            if (a.eps_free) hipLaunchKernelGGL((stft_pk16r_kernel<4, HS>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((stft_pk16r_kernel<3, HS>), grid, block, 0, stream, a);
            break;
        default: hipLaunchKernelGGL((stft_pk16r_kernel<1, HS>), grid, block, 0, stream, a); break;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}
#endif

// N = 8192 / 4096 / 2048: the same structure one, two, three sizes down (stft_pk16s.h)
template <int LOG2M, int HS>
static int launch_pk16s(const StftArgs& a, hipStream_t stream) {
    const dim3 grid(a.n_groups), block(Pk16sPlan<LOG2M>::BLOCK);
    switch (a.kind) {
        case FRT_STFT_PSD: hipLaunchKernelGGL((stft_pk16s_kernel<LOG2M, 0, HS>), grid, block, 0, stream, a); break;
        case FRT_STFT_IMAGE:
            if (a.eps_free) hipLaunchKernelGGL((stft_pk16s_kernel<LOG2M, 4, HS>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((stft_pk16s_kernel<LOG2M, 3, HS>), grid, block, 0, stream, a);
            break;
        default: hipLaunchKernelGGL((stft_pk16s_kernel<LOG2M, 1, HS>), grid, block, 0, stream, a); break;
    }
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

// ---- host side -----------------------------------------------------------------------------------

template <typename T, int LOG2M>
static int launch_big_one(const StftArgs& a, hipStream_t stream) {
    using B = BigPlan<LOG2M>;
    const int blocks = (a.n_groups + B::GPB - 1) / B::GPB;
    // rows on 16-byte boundaries (the library's own staging buffers and torch tensors are): the instances that take their
    // samples by LDS-DMA.  (-DFRT_EXPERIMENTS builds can step back a kernel generation per size: tools/exp A/B runs.)
    const bool aligned16 = ((uintptr_t)a.x % 16 == 0) && (a.x_stride % 4 == 0) && (a.hop % 4 == 0);
    if constexpr (sizeof(T) == 4) {
        if (aligned16 && !exp_env("FRT_STFT_NO_DMA")) {
            const bool half = a.hop == B::M, quarter = a.hop == B::M / 2;        // hop N/2, N/4
            if constexpr (LOG2M >= 10 && LOG2M <= 12) {
                if (!exp_env(LOG2M == 10 ? "FRT_STFT_NO_PK16W" : LOG2M == 11 ? "FRT_STFT_NO_PK16Q" : "FRT_STFT_NO_PK16H")) {
                    if (half) return launch_pk16s<LOG2M, 8>(a, stream);
                    if (quarter) return launch_pk16s<LOG2M, 4>(a, stream);
                }
            }
            if constexpr (LOG2M == Pk16Plan::LOG2M) {
#ifdef FRT_EXPERIMENTS
                if (exp_env("FRT_STFT_PK16R")) {
                    if (half) return launch_pk16r<8>(a, stream);
                    if (quarter) return launch_pk16r<4>(a, stream);
                }
#endif
                if (!exp_env("FRT_STFT_NO_PK16") && !exp_env("FRT_STFT_NO_PK")) {
                    if (half) return launch_pk16<8>(a, stream);
                    if (quarter) return launch_pk16<4>(a, stream);
                }
#ifdef FRT_EXPERIMENTS
                if (!exp_env("FRT_STFT_NO_PK")) {                 // round 3's instance (8 x 8 x 8 sub-transforms)
                    if (half) return launch_pk<8>(a, stream);
                    if (quarter) return launch_pk<4>(a, stream);
                }
#endif
            }
            if constexpr (LOG2M >= FRT_BIG_DMA_MIN_LOG2M) {
                hipLaunchKernelGGL((stft_big_kernel<T, LOG2M, true>), dim3(blocks), dim3(B::BLOCK), 0, stream, a);
                FRT_HIP_CHECK(hipGetLastError());
                return FRT_OK;
            }
        }
    }
    hipLaunchKernelGGL((stft_big_kernel<T, LOG2M, false>), dim3(blocks), dim3(B::BLOCK), 0, stream, a);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

template <typename T>
static int launch_big(int log2m, const StftArgs& a, hipStream_t stream) {
    switch (log2m) {
        case 10: return launch_big_one<T, 10>(a, stream);
        case 11: return launch_big_one<T, 11>(a, stream);
        case 12: return launch_big_one<T, 12>(a, stream);
        case 13: return launch_big_one<T, 13>(a, stream);
    }
    set_last_error("big kernel: unsupported fft size 2^%d", log2m + 1);
    return FRT_ERR_UNSUPPORTED;
}


template <typename TIN, typename T, int LOG2M, int SHIFT>
static int launch_one(const StftArgs& a, int blocks, hipStream_t stream) {
    using P = Pow2Plan<LOG2M>;
    constexpr int BLOCK = P::TPF < 256 ? 256 : P::TPF;
    if (a.out_nyq) {                            // split rows (frt_stft_run_split): N <= 1024
        if constexpr (LOG2M <= 9) {
            hipLaunchKernelGGL((stft_kernel<TIN, T, LOG2M, SHIFT, true>), dim3(blocks), dim3(BLOCK), 0, stream, a);
            FRT_HIP_CHECK(hipGetLastError());
            return FRT_OK;
        }
        set_last_error("split output rows need fft_size <= 1024");
        return FRT_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((stft_kernel<TIN, T, LOG2M, SHIFT>), dim3(blocks), dim3(BLOCK), 0, stream, a);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

template <typename TIN, typename T, int LOG2M>
static int launch_shift(const StftArgs& a, int shift, int blocks, hipStream_t stream) {
    if constexpr (sizeof(T) == 4 && LOG2M == 9) {
        if (shift == -1) return launch_one<TIN, T, LOG2M, -1>(a, blocks, stream);
    }
    if (shift < 0) shift = 4;
    // N >= 2048 comes here only off the 8-byte grid (no register window then) or through a negative run length (tests: the generic
    // workgroup walk against the radix-16 instances): the slot-reloading instance serves both — no window instances of those sizes
    // (an `else`, not a `return` in front of the rest: statements behind a constexpr-if's return are still instantiated, and the device
    // side would go on emitting their kernels)
    if constexpr (LOG2M >= 10) {
        return launch_one<TIN, T, LOG2M, 0>(a, blocks, stream);
    } else {
        if constexpr (sizeof(T) == 4) {
            if (shift == 2) return launch_one<TIN, T, LOG2M, 2>(a, blocks, stream);
        }
        // hop = N/2 keeps half of the register window (float64 too: the reference's own precision at BASELINE's overlap)
        if (shift == 4) return launch_one<TIN, T, LOG2M, 4>(a, blocks, stream);
        return launch_one<TIN, T, LOG2M, 0>(a, blocks, stream);
    }
}

template <typename TIN, typename T>
static int launch_size(int log2m, const StftArgs& a, int shift, int blocks, hipStream_t stream) {
    switch (log2m) {
        case 4: return launch_shift<TIN, T, 4>(a, shift, blocks, stream);
        case 5: return launch_shift<TIN, T, 5>(a, shift, blocks, stream);
        case 6: return launch_shift<TIN, T, 6>(a, shift, blocks, stream);
        case 7: return launch_shift<TIN, T, 7>(a, shift, blocks, stream);
        case 8: return launch_shift<TIN, T, 8>(a, shift, blocks, stream);
        case 9: return launch_shift<TIN, T, 9>(a, shift, blocks, stream);
        case 10: return launch_shift<TIN, T, 10>(a, shift, blocks, stream);
        case 11: return launch_shift<TIN, T, 11>(a, shift, blocks, stream);
        case 12: return launch_shift<TIN, T, 12>(a, shift, blocks, stream);
        case 13: return launch_shift<TIN, T, 13>(a, shift, blocks, stream);
    }
    set_last_error("unsupported fft size 2^%d", log2m + 1);
    return FRT_ERR_UNSUPPORTED;
}

}  // namespace frt

using namespace frt;

struct frt_stft {
    int fft_size = 0, hop = 0, n_channels = 0, precision = 32, log2m = 0;
    int run_length = 0;
    bool force_generic = false;   // A/B and tests: a negative run length selects the generic kernel
    hipStream_t stream = nullptr;
    DeviceBuffer window, tw, twn, tws, weight, wimage, edge_pow, bin_pow, lut;
    bool eps_free = false;
    float edge2 = 0.f;            // see exact_colour_index
    double image_thr = 0.0;
    bool has_weight = false, has_lut = false;
    double spec_min = -140.0, spec_max = 0.0;
    DeviceBuffer stage_in, stage_out;
    char* pin = nullptr;          // pinned staging of the host-buffer path: [input][output]
    size_t pin_bytes = 0;
    hipEvent_t pin_done = nullptr;     // host samples -> device spectra returns without waiting: guards the pinned block
    bool pin_pending = false;
};

template <typename T>
static int build_tables(frt_stft* h) {
    const int N = h->fft_size, M = N / 2;
    const double pi = 3.14159265358979323846;
    std::vector<T> win(N);
    // symmetric Hann, audioproc.py:76-80: 0.5 * (1 - cos(2 pi n / (N - 1)))
    // stored pre-scaled by 1/(2N): |X[k]|^2 / N^2 = |sum ... |^2 / 4 of the unpack then needs no multiply;
    // N is a power of two, so the scaling is exact and results are unchanged
    for (int n = 0; n < N; ++n) win[n] = (T)(0.5 * (1.0 - std::cos(2.0 * pi * n / (N - 1)))) * (T)(0.5 / N);
    std::vector<T> tw(2 * M), twn(2 * M);
    for (int n = 0; n < M; ++n) {
        tw[2 * n] = (T)std::cos(2.0 * pi * n / M);
        tw[2 * n + 1] = (T)(-std::sin(2.0 * pi * n / M));
        twn[2 * n] = (T)std::cos(2.0 * pi * n / N);
        twn[2 * n + 1] = (T)(-std::sin(2.0 * pi * n / N));
    }
    int rc;
    if ((rc = upload(h->window, win))) return rc;
    if ((rc = upload(h->tw, tw))) return rc;
    if ((rc = upload(h->twn, twn))) return rc;
    if (M >= 1024) {                       // sub-transform table of stft_big_kernel
        const int Ms = M / 16;
        std::vector<T> tws(2 * Ms);
        for (int n = 0; n < Ms; ++n) {
            tws[2 * n] = (T)std::cos(2.0 * pi * n / Ms);
            tws[2 * n + 1] = (T)(-std::sin(2.0 * pi * n / Ms));
        }
        if ((rc = upload(h->tws, tws))) return rc;
    }
    return FRT_OK;
}

extern "C" int frt_stft_create(frt_stft** out, int fft_size, int hop, int n_channels, int precision) {
    FRT_REQUIRE(out != nullptr, "frt_stft_create: null handle pointer");
    *out = nullptr;
    FRT_REQUIRE(fft_size >= 32 && fft_size <= 16384 && (fft_size & (fft_size - 1)) == 0,
                "frt_stft_create: fft_size %d is not a power of two in [32, 16384]", fft_size);
    FRT_REQUIRE(hop >= 1, "frt_stft_create: hop %d < 1", hop);
    FRT_REQUIRE(n_channels >= 1, "frt_stft_create: n_channels %d < 1", n_channels);
    FRT_REQUIRE(precision == 32 || precision == 64, "frt_stft_create: precision must be 32 or 64");
    frt_stft* h = new frt_stft();
    h->fft_size = fft_size;
    h->hop = hop;
    h->n_channels = n_channels;
    h->precision = precision;
    int l = 0;
    while ((2 << l) < fft_size) ++l;   // M = 2^l
    h->log2m = l;
    int rc = precision == 32 ? build_tables<float>(h) : build_tables<double>(h);
    if (rc) {
        frt_stft_destroy(h);
        return rc;
    }
    *out = h;
    return FRT_OK;
}

extern "C" void frt_stft_destroy(frt_stft* h) {
    if (!h) return;
    free_retired_allocations(true);      // blocks parked by growing buffers (common.h); synchronises the device like the releases below
    h->window.release();
    h->tw.release();
    h->twn.release();
    h->tws.release();
    h->weight.release();
    h->wimage.release();
    h->edge_pow.release();
    h->bin_pow.release();
    h->lut.release();
    h->stage_in.release();
    h->stage_out.release();
    if (h->pin_pending) (void)hipEventSynchronize(h->pin_done);
    if (h->pin_done) (void)hipEventDestroy(h->pin_done);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
}

extern "C" int frt_stft_set_stream(frt_stft* h, void* s) {
    FRT_REQUIRE(h, "frt_stft_set_stream: null handle");
    h->stream = (hipStream_t)s;
    return FRT_OK;
}

extern "C" int frt_stft_set_run_length(frt_stft* h, int r) {
    FRT_REQUIRE(h, "frt_stft_set_run_length: null handle");
    h->run_length = r < 0 ? -r : r;
    h->force_generic = r < 0;
    return FRT_OK;
}

extern "C" int frt_stft_set_epilogue(frt_stft* h, const double* weight_db, double spec_min, double spec_max,
                                     const uint32_t* lut256) {
    FRT_REQUIRE(h, "frt_stft_set_epilogue: null handle");
    FRT_REQUIRE(spec_max != spec_min, "frt_stft_set_epilogue: empty dB range");
    const int nb = h->fft_size / 2 + 1;
    int rc;
    // launches already enqueued on the handle's stream may still read the tables replaced below
    if (h->wimage.ptr) FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->has_weight = weight_db != nullptr;
    // per-bin offset of the colour index (IMAGE kind), weighting and dB range folded in
    const double span = spec_max - spec_min;
    {
        // q = fma(gain, L, wimage[k]) in float32, L = v_log_f32(P + 1e-30), against v * 255 in float64.  Error terms, in
        // index units, for a q inside the LUT's range (|q| < 256, hence |gain L| < |wimage| + 256):
        //   v_log_f32: 1 ulp of its result (ISA), |L| < 128                      gain * 2^-17
        //   P + 1e-30 rounded (or left out: eps_free, relative 2^-24.7)          gain * 1.45 * 2^-24
        //   (float64 instance: P rounded to float32 first)                       gain * 1.45 * 2^-24
        //   gain rounded to float32                                              2^-24 (|wimage| + 256)
        //   wimage[k] rounded to float32                                         2^-24 |wimage|
        //   the fma's single rounding                                            2^-24 * 256
        // thr = their sum (+5 %); the index value carries +thr (inside the float32 instance's table, added per bin by the
        // float64 instance), so the float64 value lies in (q - 2 thr, q) and the kernel re-decides exactly the bins with
        // fract(q) < 2 thr.
        const double gain = std::fabs(255.0 * 3.01029995663981195 / span);
        // (|wimage| over the bins that can reach the LUT's range at all: with |log2| < 128 a bin whose offset lies more than
        // 128 gain outside [0, 256] is clamped whatever its power — bin 0 of the A curve sits at -1000 dB — and needs no margin)
        double wmax = 0.0;
        for (int k = 0; k < nb; ++k) {
            const double wk = 255.0 * ((weight_db ? weight_db[k] : 0.0) - spec_min) / span;
            if (wk + 128.0 * gain >= 0.0 && wk - 128.0 * gain <= 256.0) wmax = std::fmax(wmax, std::fabs(wk));
        }
        const double eps24 = 1.0 / 16777216.0;
        const double thr = 1.05 * (gain * (1.0 / 131072.0 + (h->precision == 32 ? 1.5 : 3.0) * eps24) + eps24 * (2.0 * (wmax + 1.0) + 512.0));
        h->edge2 = (float)(2.0 * thr);
        h->image_thr = thr;
        std::vector<double> bp(nb), ep(256);
        double wmax_db = -1e300;
        for (int k = 0; k < nb; ++k) {
            const double wk = weight_db ? weight_db[k] : 0.0;
            bp[k] = std::pow(10.0, -wk / 10.0);
            wmax_db = std::fmax(wmax_db, wk);
        }
        for (int n = 0; n < 256; ++n) ep[n] = std::pow(10.0, (spec_min + n * (span / 255.0)) / 10.0);
        // P < 2^-75 (2.6e-23) is where P + 1e-30 differs from P in float32; if even there, with the largest weight, the
        // index value stays below 0, every such bin is clamped to index 0 with or without the 1e-30
        h->eps_free = span > 0 && (10.0 * std::log10(2.7e-23) + wmax_db - spec_min) / span * 255.0 + thr < 0.0;
        if ((rc = upload(h->edge_pow, ep)) || (rc = upload(h->bin_pow, bp))) return rc;
        if (h->precision == 32) {
            std::vector<float> w(nb), wi(nb);
            for (int k = 0; k < nb; ++k) {
                const double wk = weight_db ? weight_db[k] : 0.0;
                w[k] = (float)wk;
                wi[k] = (float)(255.0 * (wk - spec_min) / span + thr);
            }
            if ((rc = upload(h->weight, w)) || (rc = upload(h->wimage, wi))) return rc;
        } else {
            std::vector<double> w(nb), wi(nb);
            for (int k = 0; k < nb; ++k) {
                w[k] = weight_db ? weight_db[k] : 0.0;
                wi[k] = 255.0 * (w[k] - spec_min) / span;          // exact: the large-frame float64 instance evaluates in float64
            }
            if ((rc = upload(h->weight, w)) || (rc = upload(h->wimage, wi))) return rc;
        }
    }
    h->has_lut = lut256 != nullptr;
    if (lut256) {
        std::vector<uint32_t> l(lut256, lut256 + 256);
        if ((rc = upload(h->lut, l))) return rc;

    }
    h->spec_min = spec_min;
    h->spec_max = spec_max;
    return FRT_OK;
}

extern "C" int64_t frt_stft_frames_for(const frt_stft* h, int64_t T) {
    if (!h || T < h->fft_size) return 0;
    return (T - h->fft_size) / h->hop + 1;
}

static int stft_launch(frt_stft* h, int kind, const void* d_x, int64_t x_stride, void* d_out, void* d_nyq, int64_t F,
                       hipStream_t stream) {
    const int N = h->fft_size, M = N / 2;
    StftArgs a{};
    a.x = d_x;
    a.out = d_out;
    a.out_nyq = d_nyq;
    a.window = h->window.ptr;
    a.tw = h->tw.ptr;
    a.twn = h->twn.ptr;
    a.tws = h->tws.ptr;
    a.weight = h->has_weight ? h->weight.ptr : nullptr;
    a.wimage = h->wimage.ptr;
    a.image_gain = 255.0 * 3.01029995663981195 / (h->spec_max - h->spec_min);
    a.lut = h->has_lut ? h->lut.as<uint32_t>() : nullptr;
    a.x_stride = x_stride;
    a.n_frames = F;
    a.out_cstride = F * (d_nyq ? M : M + 1);
    a.hop = h->hop;
    a.kind = kind;
    const size_t esz = h->precision == 32 ? 4 : 8;
    a.vec2 = (h->hop % 2 == 0) && (x_stride % 2 == 0) && (((uintptr_t)d_x) % (2 * esz) == 0);
    a.norm_off = -h->spec_min;
    a.norm_scale = 1.0 / (h->spec_max - h->spec_min);
    a.edge_pow = h->edge_pow.as<double>();
    a.bin_pow = h->bin_pow.as<double>();
    a.edge2 = h->edge2;
    a.image_thr = h->image_thr;
#ifdef FRT_EXPERIMENTS
    if (const char* e = exp_env("FRT_EDGE_SCALE")) a.edge2 *= (float)atof(e);      // tools/exp: how wide must the float64 zone be?
#endif
    a.rising = h->spec_max > h->spec_min;
    a.eps_free = h->eps_free;
#ifdef FRT_ABLATE
    a.ablate = exp_int("FRT_ABLATE", 0);        // (-DFRT_ABLATE builds are experiment builds: add -DFRT_EXPERIMENTS)
#endif

    // slots of 2*TPF samples a hop advances; the register-shift kernels need hop = s*N/8, s in {2,4}
    int shift = 0;
    if (a.vec2 && h->hop * 8 % N == 0) {
        const int s = h->hop * 8 / N;
        if (s == 2 || s == 4) shift = s;
    }
    // N = 1024, hop 512, rows on 16-byte boundaries: the ring instance (samples through a per-wavefront LDS ring filled by LDS-DMA)
    static const bool no_ring = exp_env("FRT_STFT_NO_RING") != nullptr;       // A/B runs: the register-window instance
    // (PSD / dB kinds: +4 % over the register window, 60-61 % of HBM peak; the colour kind, whose epilogue adds its own LDS
    // gathers, measures equal or 0.5 % behind and keeps the register window unless FRT_STFT_RING_IMAGE is set)
    static const bool ring_image = exp_env("FRT_STFT_RING_IMAGE") != nullptr;
    if (shift == 4 && h->log2m == 9 && h->precision == 32 && ((uintptr_t)d_x % 16 == 0) && (x_stride % 4 == 0) && !no_ring &&
        (kind != FRT_STFT_IMAGE || ring_image))
        shift = -1;
    const int tpf = M / 8;
    const int gpb = tpf < 256 ? 256 / tpf : 1;
    // run length: enough lane groups to fill the chip several times over, long enough to amortise
    // the one-off loads of a run (window, twiddles, first frame)
    int run = h->run_length;
    if (run <= 0) {
        const long long total = (long long)F * h->n_channels;
        const long long want_groups = (long long)device_cu_count() * 8 * gpb;
        run = (int)((total + want_groups - 1) / want_groups);
        if (run < 8) run = 8;
        if (run > 64) run = 64;
    }
    if (run > F) run = (int)F;
    if (d_nyq && run > 64) run = 64;            // a run's Nyquist values ride in one register, a lane per frame
    a.run = run;
    a.frame_base = 0;

    // N >= 2048, aligned even hop: the radix-16 + wave-local instances (stft_big.h)
    const bool big_ok = h->log2m >= 10;
    FRT_REQUIRE_CODE(!(big_ok && d_nyq), FRT_ERR_UNSUPPORTED, "frt_stft_run_split: split output rows need fft_size <= 1024 (got %d)", N);
    if (big_ok && a.vec2 && !h->force_generic) {
        int brun = h->run_length;
        if (brun <= 0) {
            // every thread keeps its window, twiddle and weight factors in registers for the whole run (94 table loads
            // against 16 sample loads per frame), so runs should be long — as long as the groups still fill the chip.
            // Measured (run sweep with tools/stft_selftest bench, one frame per workgroup): 8 frames is the plateau at
            // N = 4096 / 8192, 16 at N = 16384; N = 2048 keeps gaining up to 32 (+6 % over 8) provided two full
            // rounds of groups remain.
            // (N = 16384: one; two for -DFRT_EXPERIMENTS runs of stft_pk16r_kernel)
            const bool two_per_cu = h->precision == 32 && exp_env("FRT_STFT_PK16R") != nullptr;
            const int resident = h->log2m >= 13 ? (two_per_cu ? 2 : 1) : h->log2m == 12 ? 2 : h->log2m == 11 ? 4 : 8;   // groups per CU
            const long long need = (long long)device_cu_count() * resident;
            auto groups = [&](int r) { return ((F + r - 1) / r) * h->n_channels; };
            brun = h->log2m >= 10 ? 16 : 8;
            while (brun > 1 && groups(brun) < need) brun /= 2;
            // (only when the batch fills the chip at all: a widget-sized or catch-up call of a few frames keeps the halved run —
            // four frames of one channel are four workgroups side by side, not one workgroup walking them one after the other)
            if (h->log2m >= 10 && groups(brun) >= need) {
                // one workgroup per CU (N = 16384), two (N = 8192), four (N = 4096) or eight (N = 2048): the groups should come in whole rounds of the chip (F = 253 frames x 32 channels in
                // runs of 16 are 512 groups = two rounds, the second one short; in runs of 32 one round) with runs as long as
                // that allows (every run re-reads N - hop samples of its predecessor and loads ~120 constants per thread)
                const long long total = (long long)F * h->n_channels;
                const int cap = h->log2m == 10 ? 16 : 48;                                     // frames per run at most (measured per size)
                const long long rounds = (total + need * cap - 1) / (need * cap);
                long long rpc = (need * rounds + h->n_channels / 2) / h->n_channels;          // runs per channel
                if (rpc < 1) rpc = 1;
                long long r = (F + rpc - 1) / rpc;
                const int rmin = h->log2m == 10 ? 12 : 8;       // (N = 2048, one channel of 2^24 samples: runs of 8 / 12 / 16: 0.45 / 0.50 / 0.49)
                if (r < rmin) r = rmin;
                if (r <= 64) brun = (int)r;
            }
        }
        if (brun > F) brun = (int)F;
        a.run = brun;
        a.runs_per_channel = (int)((F + brun - 1) / brun);
        const long long bgroups = (long long)a.runs_per_channel * h->n_channels;
        FRT_REQUIRE(bgroups < (1ll << 31), "frt_stft_run: too many lane groups");
        a.n_groups = (int)bgroups;
        return h->precision == 32 ? launch_big<float>(h->log2m, a, stream) : launch_big<double>(h->log2m, a, stream);
    }
    const long long rest = F - a.frame_base;
    a.runs_per_channel = (int)((rest + run - 1) / run);
    const long long groups = (long long)a.runs_per_channel * h->n_channels;
    FRT_REQUIRE(groups < (1ll << 31), "frt_stft_run: too many lane groups");
    a.n_groups = (int)groups;
    const int blocks = (int)((groups + gpb - 1) / gpb);
    if (h->precision == 32) return launch_size<float, float>(h->log2m, a, shift, blocks, stream);
    return launch_size<double, double>(h->log2m, a, shift, blocks, stream);
}

static int stft_run(frt_stft* h, int kind, const void* x, int64_t T, int64_t x_stride, void* out, void* nyq, bool split,
                    int64_t* n_frames_out) {
    FRT_REQUIRE(h, "frt_stft_run: null handle");
    FRT_REQUIRE(kind >= FRT_STFT_PSD && kind <= FRT_STFT_IMAGE, "frt_stft_run: unknown output kind %d", kind);
    FRT_REQUIRE(kind != FRT_STFT_IMAGE || h->has_lut, "frt_stft_run: IMAGE output needs a colour LUT");
    FRT_REQUIRE(T >= 0 && x_stride >= T, "frt_stft_run: bad T/x_stride");
    const int64_t F = frt_stft_frames_for(h, T);
    if (n_frames_out) *n_frames_out = F;
    if (F == 0) return FRT_OK;
    FRT_REQUIRE(x && out, "frt_stft_run: null buffer");
    FRT_REQUIRE(!split || nyq, "frt_stft_run_split: null Nyquist plane");
    const bool dx = is_device_pointer(x), dout = is_device_pointer(out);
    FRT_REQUIRE(dx == dout || dout, "frt_stft_run: device samples need a device output");
    FRT_REQUIRE(!split || is_device_pointer(nyq) == dout, "frt_stft_run_split: rows and Nyquist plane must live on the same side");
    const int nb = split ? h->fft_size / 2 : h->fft_size / 2 + 1;        // values per row
    const size_t in_esz = h->precision == 32 ? 4 : 8;
    const size_t out_esz = (kind == FRT_STFT_IMAGE) ? 4 : in_esz;
    if (dx) return stft_launch(h, kind, x, x_stride, out, nyq, F, h->stream);
    if (h->pin_pending) {                        // an earlier host -> device call may still be reading the pinned block
        FRT_HIP_CHECK(hipEventSynchronize(h->pin_done));
        h->pin_pending = false;
    }
    if (dout) {
        // host samples, spectra that stay on the device (a widget's ring on the host, its read-out chain on the device):
        // the samples go through the pinned block — read in place by the kernel when small — and the call returns without
        // waiting; consumers order themselves on the handle's stream
        const size_t in_bytes = (size_t)h->n_channels * x_stride * in_esz;
        int rc;
        if (in_bytes > h->pin_bytes) {
            FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
            if (h->pin) (void)hipHostFree(h->pin);
            h->pin = nullptr;
            h->pin_bytes = 0;
            FRT_HIP_CHECK(hipHostMalloc((void**)&h->pin, 2 * in_bytes, hipHostMallocDefault));
            h->pin_bytes = 2 * in_bytes;
        }
        memcpy(h->pin, x, in_bytes);
        const void* src = h->pin;
        if (in_bytes > kZeroCopyMax) {
            if ((rc = h->stage_in.reserve(in_bytes))) return rc;
            FRT_HIP_CHECK(hipMemcpyAsync(h->stage_in.ptr, h->pin, in_bytes, hipMemcpyHostToDevice, h->stream));
            src = h->stage_in.ptr;
        }
        if ((rc = stft_launch(h, kind, src, x_stride, out, nyq, F, h->stream))) return rc;
        if (!h->pin_done) FRT_HIP_CHECK(hipEventCreateWithFlags(&h->pin_done, hipEventDisableTiming));
        FRT_HIP_CHECK(hipEventRecord(h->pin_done, h->stream));
        h->pin_pending = true;
        return FRT_OK;
    }

    // host buffers: stage through device memory, return when the result is back.  Small calls (the widgets' one frame at a
    // time: audioproc.analyzelive) go through the handle's pinned block — from pageable memory the runtime stages every
    // copy itself and blocks the caller twice; large ones are copied in place.
    const size_t in_bytes = (size_t)h->n_channels * x_stride * in_esz;
    const size_t row_bytes = (size_t)h->n_channels * F * nb * out_esz;                       // split: the Nyquist plane follows the rows
    const size_t out_bytes = row_bytes + (split ? (size_t)h->n_channels * F * out_esz : 0);
    int rc;
    if ((rc = h->stage_in.reserve(in_bytes))) return rc;
    if ((rc = h->stage_out.reserve(out_bytes))) return rc;
    const size_t in_pad = (in_bytes + 255) / 256 * 256;
    const bool pinned = in_pad + out_bytes <= (size_t)1 << 22;
    if (pinned && in_pad + out_bytes > h->pin_bytes) {
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr;
        h->pin_bytes = 0;
        FRT_HIP_CHECK(hipHostMalloc((void**)&h->pin, 2 * (in_pad + out_bytes), hipHostMallocDefault));
        h->pin_bytes = 2 * (in_pad + out_bytes);
    }
    if (pinned) memcpy(h->pin, x, in_bytes);
    if (pinned && in_pad + out_bytes <= kZeroCopyMax) {
        // one frame or a few (audioproc.analyzelive): the kernel reads the pinned block and writes the spectrum into it — no
        // copy engine on either side, one launch and one synchronisation per call
        if ((rc = stft_launch(h, kind, h->pin, x_stride, h->pin + in_pad, split ? h->pin + in_pad + row_bytes : nullptr, F, h->stream))) return rc;
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        memcpy(out, h->pin + in_pad, row_bytes);
        if (split) memcpy(nyq, h->pin + in_pad + row_bytes, out_bytes - row_bytes);
        return FRT_OK;
    }
    FRT_HIP_CHECK(hipMemcpyAsync(h->stage_in.ptr, pinned ? (const void*)h->pin : x, in_bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = stft_launch(h, kind, h->stage_in.ptr, x_stride, h->stage_out.ptr, split ? (char*)h->stage_out.ptr + row_bytes : nullptr, F, h->stream))) return rc;
    if (pinned) {
        FRT_HIP_CHECK(hipMemcpyAsync(h->pin + in_pad, h->stage_out.ptr, out_bytes, hipMemcpyDeviceToHost, h->stream));
    } else {
        FRT_HIP_CHECK(hipMemcpyAsync(out, h->stage_out.ptr, row_bytes, hipMemcpyDeviceToHost, h->stream));
        if (split) FRT_HIP_CHECK(hipMemcpyAsync(nyq, (char*)h->stage_out.ptr + row_bytes, out_bytes - row_bytes, hipMemcpyDeviceToHost, h->stream));
    }
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (pinned) {
        memcpy(out, h->pin + in_pad, row_bytes);
        if (split) memcpy(nyq, h->pin + in_pad + row_bytes, out_bytes - row_bytes);
    }
    return FRT_OK;
}

extern "C" int frt_stft_run(frt_stft* h, int kind, const void* x, int64_t T, int64_t x_stride, void* out,
                            int64_t* n_frames_out) {
    return stft_run(h, kind, x, T, x_stride, out, nullptr, false, n_frames_out);
}

extern "C" int frt_stft_run_split(frt_stft* h, int kind, const void* x, int64_t T, int64_t x_stride, void* out_rows,
                                  void* out_nyquist, int64_t* n_frames_out) {
    FRT_REQUIRE(h, "frt_stft_run_split: null handle");
    FRT_REQUIRE_CODE(h->fft_size <= 1024, FRT_ERR_UNSUPPORTED, "frt_stft_run_split: split output rows need fft_size <= 1024 (got %d)", h->fft_size);
    return stft_run(h, kind, x, T, x_stride, out_rows, out_nyquist, true, n_frames_out);
}

extern "C" int frt_stft_psd(frt_stft* h, const float* x, int64_t T, float* psd_out, int64_t* n_frames_out) {
    FRT_REQUIRE(h && h->precision == 32, "frt_stft_psd: needs a precision-32 handle");
    return frt_stft_run(h, FRT_STFT_PSD, x, T, T, psd_out, n_frames_out);
}

extern "C" int frt_stft_image(frt_stft* h, const float* x, int64_t T, uint32_t* rgba_out, int64_t* n_frames_out) {
    FRT_REQUIRE(h && h->precision == 32, "frt_stft_image: needs a precision-32 handle");
    return frt_stft_run(h, FRT_STFT_IMAGE, x, T, T, rgba_out, n_frames_out);
}

extern "C" int frt_stft_analyzelive_f64(frt_stft* h, const double* frame, double* psd_out) {
    FRT_REQUIRE(h && h->precision == 64, "frt_stft_analyzelive_f64: needs a precision-64 handle");
    FRT_REQUIRE(h->n_channels == 1, "frt_stft_analyzelive_f64: handle must have one channel");
    int64_t F = 0;
    int rc = frt_stft_run(h, FRT_STFT_PSD, frame, h->fft_size, h->fft_size, psd_out, &F);
    if (rc) return rc;
    FRT_REQUIRE(F == 1, "frt_stft_analyzelive_f64: internal frame count %lld", (long long)F);
    return FRT_OK;
}
