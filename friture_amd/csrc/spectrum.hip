// spectrum.hip — post-processing of the spectrum widget on the device, gfx950.
//
// Reference semantics: Spectrum_Widget.handle_new_data, friture/spectrum.py:156-182
//   sp   = exp_smoothed_value_2d(kernel, alpha, spn, previous)      (spn: (bins, frames))
//   dB   = 10 log10(sp + 1e-30) + w            (dual channel: 10 log10(sp2 + eps) - 10 log10(sp1 + eps))
//   peak = argmax(dB)
//   hps  = sp[:K] * sp[::2][:K] * sp[::3][:K],  K = bins // 3;  pitch = argmax(hps)   (:103-123)
// One workgroup handles one channel: the PSD frames of a call are a few hundred KB at most, the work
// is a strided reduction per bin followed by two arg-max reductions — fused so that the PSD slab the
// STFT kernel left in HBM is read once and only bins-sized vectors travel back.
#include <cmath>

#include "common.h"

namespace frt {

constexpr int kPostThreads = 1024;

struct ArgMax {
    double v;
    int i;
};

__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {          // first index wins ties (numpy.argmax)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__device__ ArgMax block_argmax(ArgMax m, double* red, int* redi) {
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax other = {__shfl_down(m.v, o, 64), __shfl_down(m.i, o, 64)};
        m = better(m, other);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = m.v;
        redi[threadIdx.x >> 6] = m.i;
    }
    __syncthreads();
    ArgMax r = {red[0], redi[0]};
    for (int w = 1; w < kPostThreads / 64; ++w) r = better(r, ArgMax{red[w], redi[w]});
    return r;
}

template <typename TP>
__global__ void __launch_bounds__(kPostThreads) spectrum_post_kernel(
    const TP* __restrict__ psd, long long frame_stride, int nt, int n_bins, const double* __restrict__ kern, double alpha,
    double decay, const double* __restrict__ previous, const double* __restrict__ weight, const double* __restrict__ ref,
    double* __restrict__ smoothed, double* __restrict__ db, int* __restrict__ idx_out) {
    __shared__ double red[kPostThreads / 64];
    __shared__ int redi[kPostThreads / 64];
    const int tid = threadIdx.x;
    ArgMax best = {-INFINITY, 0x7fffffff};
    for (int k = tid; k < n_bins; k += kPostThreads) {
        double acc = 0.0;
        for (int t = 0; t < nt; ++t) acc += (double)psd[(long long)t * frame_stride + k] * kern[t];
        const double sp = alpha * acc + previous[k] * decay;
        smoothed[k] = sp;
        double d = 10.0 * log10(sp + 1e-30);
        if (ref) d -= 10.0 * log10(ref[k] + 1e-30);
        else if (weight) d += weight[k];
        db[k] = d;
        if (!(d != d)) best = better(best, ArgMax{d, k});
    }
    best = block_argmax(best, red, redi);
    if (tid == 0) idx_out[0] = best.i == 0x7fffffff ? 0 : best.i;
    __threadfence_block();
    __syncthreads();
    // harmonic product spectrum on the smoothed spectrum (or on `ref`, the first channel, in dual mode)
    const double* s = ref ? ref : smoothed;
    const int K = n_bins / 3;
    ArgMax hb = {-INFINITY, 0x7fffffff};
    for (int k = tid; k < K; k += kPostThreads) {
        const double h = s[k] * s[2 * k] * s[3 * k];
        if (!(h != h)) hb = better(hb, ArgMax{h, k});
    }
    hb = block_argmax(hb, red, redi);
    if (tid == 0) idx_out[1] = hb.i == 0x7fffffff ? 0 : hb.i;
}

}  // namespace frt

using namespace frt;

extern "C" int frt_spectrum_post(const void* psd, int psd_is_f32, int n_frames, int n_bins, int64_t frame_stride,
                                 const double* kernel, int nk, double alpha, const double* previous, const double* weight_db,
                                 const double* ref_smoothed, double* smoothed_out, double* db_out, int* peak_index_out,
                                 int* pitch_index_out) {
    FRT_REQUIRE(n_frames >= 0 && n_bins >= 3 && frame_stride >= n_bins && nk >= 0, "frt_spectrum_post: bad sizes");
    FRT_REQUIRE(kernel && previous && smoothed_out && db_out && (n_frames == 0 || psd), "frt_spectrum_post: null buffer");
    FRT_REQUIRE(!is_device_pointer(kernel), "frt_spectrum_post: kernel is a host table");
    int n = n_frames;                           // exp_smoothing.py:94-101
    double decay;
    if (n > nk) {
        n = nk;
        decay = 0.0;
    } else {
        decay = std::pow(1.0 - alpha, (double)n);
    }
    const bool dev = is_device_pointer(smoothed_out);
    // the smoothed spectra and what they are formed from live together (all host or all device); the dB vector may come
    // back to a host array while the state stays on the device (the spectrum widget's object)
    FRT_REQUIRE(dev == is_device_pointer(previous) && (n_frames == 0 || dev == is_device_pointer(psd)) &&
                    (!weight_db || dev == is_device_pointer(weight_db)) && (!ref_smoothed || dev == is_device_pointer(ref_smoothed)) &&
                    (dev || !is_device_pointer(db_out)),
                "frt_spectrum_post: spectra, state and weights must all be host or all be device memory");
    const size_t esz = psd_is_f32 ? 4 : 8;
    // all host arguments in one pinned block, one upload, one download (StageCall, common.h): seven blocking copies and a
    // thread-local set of device buffers until round 3
    StageCall st;
    const size_t pbytes = n_frames ? ((size_t)(n_frames - 1) * frame_stride + n_bins) * esz : 0;
    const size_t vbytes = (size_t)n_bins * sizeof(double);
    const double zero = 0.0;
    const int i_k = n > 0 ? st.add_in(kernel + (nk - n), (size_t)n * sizeof(double)) : st.add_in(&zero, sizeof(double));
    const int i_psd = pbytes ? st.add_in(psd, pbytes) : -1;
    const int i_prev = st.add_in(previous, vbytes);
    const int i_w = weight_db ? st.add_in(weight_db, vbytes) : -1;
    const int i_ref = ref_smoothed ? st.add_in(ref_smoothed, vbytes) : -1;
    const int i_sm = st.add_out(smoothed_out, vbytes), i_db = st.add_out(db_out, vbytes);
    int idx[2] = {0, 0};
    const int i_idx = st.add_out(idx, sizeof(idx));
    int rc;
    if ((rc = st.begin())) return rc;
    const double* d_w = i_w >= 0 ? st.ptr<const double>(i_w) : nullptr;
    const double* d_ref = i_ref >= 0 ? st.ptr<const double>(i_ref) : nullptr;
    if (psd_is_f32)
        hipLaunchKernelGGL(spectrum_post_kernel<float>, dim3(1), dim3(kPostThreads), 0, st.stream(),
                           i_psd >= 0 ? st.ptr<const float>(i_psd) : nullptr, (long long)frame_stride, n, n_bins, st.ptr<const double>(i_k),
                           alpha, decay, st.ptr<const double>(i_prev), d_w, d_ref, st.ptr<double>(i_sm), st.ptr<double>(i_db), st.ptr<int>(i_idx));
    else
        hipLaunchKernelGGL(spectrum_post_kernel<double>, dim3(1), dim3(kPostThreads), 0, st.stream(),
                           i_psd >= 0 ? st.ptr<const double>(i_psd) : nullptr, (long long)frame_stride, n, n_bins, st.ptr<const double>(i_k),
                           alpha, decay, st.ptr<const double>(i_prev), d_w, d_ref, st.ptr<double>(i_sm), st.ptr<double>(i_db), st.ptr<int>(i_idx));
    if ((rc = st.finish())) return rc;
    if (peak_index_out) *peak_index_out = idx[0];
    if (pitch_index_out) *pitch_index_out = idx[1];
    return FRT_OK;
}
