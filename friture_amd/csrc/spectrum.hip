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
    FRT_REQUIRE(dev == is_device_pointer(db_out) && dev == is_device_pointer(previous) &&
                    (n_frames == 0 || dev == is_device_pointer(psd)) && (!weight_db || dev == is_device_pointer(weight_db)) &&
                    (!ref_smoothed || dev == is_device_pointer(ref_smoothed)),
                "frt_spectrum_post: buffers must all be host or all be device memory");
    const size_t esz = psd_is_f32 ? 4 : 8;
    // device scratch of this stateless entry point: grow-only, per calling thread (it used to be allocated and freed on
    // every call — the spectrum widget calls once per audio chunk)
    struct Scratch {
        DeviceBuffer psd, prev, w, ref, sm, db, k, idx;
        ~Scratch() { for (DeviceBuffer* b : {&psd, &prev, &w, &ref, &sm, &db, &k, &idx}) b->release(); }
    };
    static thread_local Scratch scratch;
    DeviceBuffer &b_psd = scratch.psd, &b_prev = scratch.prev, &b_w = scratch.w, &b_ref = scratch.ref, &b_sm = scratch.sm,
                 &b_db = scratch.db, &b_k = scratch.k, &b_idx = scratch.idx;
    auto release = [&]() {};
    const void* d_psd = psd;
    const double *d_prev = previous, *d_w = weight_db, *d_ref = ref_smoothed;
    double *d_sm = smoothed_out, *d_db = db_out;
    int rc = FRT_OK;
    hipError_t e = hipSuccess;
    std::vector<double> ktail(kernel + (nk - n), kernel + nk);
    if (ktail.empty()) ktail.push_back(0.0);
    if ((rc = upload(b_k, ktail)) || (rc = b_idx.reserve(2 * sizeof(int)))) { release(); return rc; }
    if (!dev) {
        const size_t pbytes = n_frames ? ((size_t)(n_frames - 1) * frame_stride + n_bins) * esz : 0;
        if ((pbytes && (rc = b_psd.reserve(pbytes))) || (rc = b_prev.reserve(n_bins * 8)) || (rc = b_sm.reserve(n_bins * 8)) ||
            (rc = b_db.reserve(n_bins * 8)) || (weight_db && (rc = b_w.reserve(n_bins * 8))) ||
            (ref_smoothed && (rc = b_ref.reserve(n_bins * 8)))) { release(); return rc; }
        if (pbytes) e = hipMemcpy(b_psd.ptr, psd, pbytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b_prev.ptr, previous, n_bins * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && weight_db) e = hipMemcpy(b_w.ptr, weight_db, n_bins * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && ref_smoothed) e = hipMemcpy(b_ref.ptr, ref_smoothed, n_bins * 8, hipMemcpyHostToDevice);
        d_psd = b_psd.ptr;
        d_prev = b_prev.as<double>();
        d_w = weight_db ? b_w.as<double>() : nullptr;
        d_ref = ref_smoothed ? b_ref.as<double>() : nullptr;
        d_sm = b_sm.as<double>();
        d_db = b_db.as<double>();
    }
    if (e == hipSuccess) {
        if (psd_is_f32)
            hipLaunchKernelGGL(spectrum_post_kernel<float>, dim3(1), dim3(kPostThreads), 0, nullptr, (const float*)d_psd,
                               (long long)frame_stride, n, n_bins, b_k.as<double>(), alpha, decay, d_prev, d_w, d_ref, d_sm, d_db,
                               b_idx.as<int>());
        else
            hipLaunchKernelGGL(spectrum_post_kernel<double>, dim3(1), dim3(kPostThreads), 0, nullptr, (const double*)d_psd,
                               (long long)frame_stride, n, n_bins, b_k.as<double>(), alpha, decay, d_prev, d_w, d_ref, d_sm, d_db,
                               b_idx.as<int>());
        e = hipGetLastError();
    }
    int idx[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(idx, b_idx.ptr, sizeof(idx), hipMemcpyDeviceToHost);
    if (e == hipSuccess && !dev) e = hipMemcpy(smoothed_out, d_sm, n_bins * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && !dev) e = hipMemcpy(db_out, d_db, n_bins * 8, hipMemcpyDeviceToHost);
    release();
    if (e != hipSuccess) {
        set_last_error("frt_spectrum_post: %s", hipGetErrorString(e));
        return FRT_ERR_HIP;
    }
    if (peak_index_out) *peak_index_out = idx[0];
    if (pitch_index_out) *pitch_index_out = idx[1];
    return FRT_OK;
}
