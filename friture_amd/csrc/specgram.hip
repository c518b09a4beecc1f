// specgram.hip — the spectrogram widget's whole chunk handler as ONE device-resident object.
//
// Reference semantics, per chunk of new samples (all float64):
//   ring push                                              friture/ringbuffer.py:39-63
//   realizable = floor(available / needed) frames ending at old_index + i * int(needed), each
//     analyzelive -> column of spn                         friture/spectrogram.py:133-159, friture/audioproc.py:42-50
//   norm = (10 log10(spn + 1e-30) + w - min) / (max - min) friture/spectrogram.py:119-129,161-162
//   Frequency_Resampler.push: np.interp per column         friture/signal/frequency_resampler.py:67-83
//   Online_Linear_2D_resampler.push (set_height first)     friture/signal/online_linear_2D_resampler.py:45-97
//   Color_Transform.push: lut[int(clip(v,0,1)*255)]        friture/signal/color_tranform.py:48-51
//   addData: frequency axis flipped, rows of Format_RGB32  friture/spectrogram_image.py:82-92,119-129
//
// What lives where.  The host keeps the integers and the three doubles the reference keeps (ring offset, old_index,
// orig_index / resampled_index / ratio: they decide HOW MANY pixel columns a chunk produces and with which weights — a
// scalar recurrence, identical here); the device keeps everything that has a length: the mirror ring, the spectra, the
// frequency map, the carried (frequency-resampled) column of the time resampler, the LUT and the pixel block.  A push is
//   H2D of the new samples only -> ring_write_kernel (both halves of the mirror) -> stft_kernel (float64 instance, NORM
//   epilogue, frames read in place from the ring: the mirror makes every window contiguous) -> screen_columns_kernel
//   (np.interp gather, time lerp against the carried column, clip, LUT, flipped rows) -> D2H of the new pixel columns.
// The three screen stages are fused but keep the reference's operations and their order (this file is compiled with
// -ffp-contract=off like pipeline.hip), so the pixels are the reference's pixels.
#include <cmath>

#include "common.h"

namespace frt {

__global__ void __launch_bounds__(256) ring_write_kernel(const double* __restrict__ chunk, int n, double* __restrict__ ring,
                                                         long long ring_len, long long offset) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const long long p = (offset + t) % ring_len;
    const double v = chunk[t];
    ring[p] = v;                       // ringbuffer.py:52-59: the second copy makes [p, p + L) a linear view of the last L samples
    ring[p + ring_len] = v;
}

// Growth of the mirror ring (ringbuffer.py:102-130 grows by x1.5 and re-lays the data): the last `keep` samples, absolute
// indices [offset - keep, offset), move from their places in the old ring to their places in the new one, both copies.
__global__ void __launch_bounds__(256) ring_relay_kernel(const double* __restrict__ old_ring, long long old_len,
                                                         double* __restrict__ new_ring, long long new_len, long long offset,
                                                         long long keep) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= keep) return;
    const long long a = offset - keep + t;
    const double v = old_ring[a % old_len];
    const long long p = a % new_len;
    new_ring[p] = v;
    new_ring[p + new_len] = v;
}

// One thread per (pixel column p, screen row h).  norm: [frames][nb] (frame-major, what stft_kernel writes).
// np.interp with the interval index found on the host (frequency_resampler.py:80; same branches as freq_resample_kernel).
__device__ __forceinline__ double freq_interp(const double* __restrict__ col, int nb, int j, double dx, double den) {
    if (j < 0) return col[0];
    if (j >= nb - 1) return col[nb - 1];
    const double f0 = col[j];
    if (dx == 0.0) return f0;
    const double slope = (col[j + 1] - f0) / den;
    return slope * dx + f0;
}

// A chunk rarely yields more than a handful of pixel columns: their source frame and weight then travel in the kernel
// arguments (no separate upload in front of the launch).
constexpr int kInlineCols = 24;
struct ColumnTable {
    int src[kInlineCols];
    double a[kInlineCols];
};

__global__ void __launch_bounds__(64) screen_columns_kernel(const double* __restrict__ norm, int nb, int n_frames,
                                                            const int* __restrict__ jidx, const double* __restrict__ dxs,
                                                            const double* __restrict__ dens, int height,
                                                            const double* __restrict__ old_in, double* __restrict__ old_out,
                                                            const int* __restrict__ src_tab, const double* __restrict__ a_tab,
                                                            const ColumnTable inl, int n_out,
                                                            const uint32_t* __restrict__ lut, uint32_t* __restrict__ pixels,
                                                            int pixel_stride, int flip) {
    const int* src = src_tab ? src_tab : inl.src;
    const double* a = a_tab ? a_tab : inl.a;
    const int p = blockIdx.x * 64 + threadIdx.x;
    const int h = blockIdx.y;
    const int j = jidx[h];
    const double dx = dxs[h], den = dens[h];
    if (p == 0) old_out[h] = freq_interp(norm + (size_t)(n_frames - 1) * nb, nb, j, dx, den);   // the column carried to the next push
    if (p >= n_out) return;
    const int c = src[p];
    const double cur = freq_interp(norm + (size_t)c * nb, nb, j, dx, den);
    const double prev = c == 0 ? old_in[h] : freq_interp(norm + (size_t)(c - 1) * nb, nb, j, dx, den);
    const double w = a[p];
    double v = cur * (1.0 - w) + prev * w;               // linear_interp.py:57-60
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);             // numpy.clip (NaN falls through to the cast like numpy's)
    pixels[(size_t)(flip ? height - 1 - h : h) * pixel_stride + p] = lut[(int)(v * 255.0)];
}

// numpy.interp's interval search for the screen rows (largest j with freq[j] <= x; -1 / nb outside the table)
static void interval_search(const double* freq, int nb, const double* targets, int height, int* j, double* dx, double* den) {
    for (int r = 0; r < height; ++r) {
        const double x = targets[r];
        dx[r] = 0.0;
        den[r] = 1.0;
        if (!(x >= freq[0])) { j[r] = -1; continue; }
        if (x > freq[nb - 1]) { j[r] = nb; continue; }
        int lo = 0, hi = nb;
        while (hi - lo > 1) {
            const int mid = (lo + hi) / 2;
            if (freq[mid] <= x) lo = mid; else hi = mid;
        }
        j[r] = lo;
        if (lo < nb - 1) {
            dx[r] = x - freq[lo];
            den[r] = freq[lo + 1] - freq[lo];
        }
    }
}

}  // namespace frt

using namespace frt;

// The three blocks of the spectrogram's Transform_Pipeline (friture/spectrogram.py:62-68: Frequency_Resampler ->
// Online_Linear_2D_resampler -> Color_Transform) as ONE call on host arrays: what Transform_Pipeline.push of the drop-in
// package does when its blocks are those three (friture_amd/signal/transform_pipeline.py) — three staged calls become one.
// norm: [n_cols][nb] (frame-major: column c of the reference's (bins, columns) block is row c); old_in: the time resampler's
// carried column (already on the screen rows); src / a: its (source column, weight) pairs for the n_out pixel columns of this
// push (the scalar index recurrence stays with the caller, as in the reference).  pixels_out: [height][n_out], row 0 =
// lowest frequency like the reference's block (the widget flips later); old_out: the last column on the screen rows.
extern "C" int frt_screen_columns(const double* norm, int nb, int n_cols, const double* freq, const double* targets, int height,
                                  const double* old_in, const int* src, const double* a, int n_out, const uint32_t* lut256,
                                  uint32_t* pixels_out, double* old_out) {
    FRT_REQUIRE(norm && freq && targets && old_in && lut256 && old_out && nb >= 1 && n_cols >= 1 && height >= 1 && n_out >= 0,
                "frt_screen_columns: bad arguments");
    FRT_REQUIRE(n_out == 0 || (src && a && pixels_out), "frt_screen_columns: null buffer");
    for (int p = 0; p < n_out; ++p) FRT_REQUIRE(src[p] >= 0 && src[p] < n_cols, "frt_screen_columns: source column out of range");
    std::vector<int> j(height);
    std::vector<double> dx(height), den(height);
    interval_search(freq, nb, targets, height, j.data(), dx.data(), den.data());
    const int cols_alloc = n_out > 0 ? n_out : 1;
    StageCall st;
    const int i_norm = st.add_in(norm, (size_t)n_cols * nb * sizeof(double)), i_j = st.add_in(j.data(), (size_t)height * sizeof(int)),
              i_dx = st.add_in(dx.data(), (size_t)height * sizeof(double)), i_den = st.add_in(den.data(), (size_t)height * sizeof(double)),
              i_old = st.add_in(old_in, (size_t)height * sizeof(double)), i_lut = st.add_in(lut256, 256 * sizeof(uint32_t)),
              i_src = st.add_in(src ? src : j.data(), (size_t)cols_alloc * sizeof(int)),
              i_a = st.add_in(a ? a : dx.data(), (size_t)cols_alloc * sizeof(double)),
              i_pix = n_out > 0 ? st.add_out(pixels_out, (size_t)height * n_out * sizeof(uint32_t)) : st.add_scratch((size_t)height * sizeof(uint32_t)),
              i_oldo = st.add_out(old_out, (size_t)height * sizeof(double));
    int rc;
    if ((rc = st.begin())) return rc;
    ColumnTable none{};
    hipLaunchKernelGGL(screen_columns_kernel, dim3((cols_alloc + 63) / 64, height), dim3(64), 0, st.stream(), st.ptr<const double>(i_norm), nb,
                       n_cols, st.ptr<const int>(i_j), st.ptr<const double>(i_dx), st.ptr<const double>(i_den), height,
                       st.ptr<const double>(i_old), st.ptr<double>(i_oldo), st.ptr<const int>(i_src), st.ptr<const double>(i_a), none, n_out,
                       st.ptr<const uint32_t>(i_lut), st.ptr<uint32_t>(i_pix), cols_alloc, 0);
    return st.finish();
}

struct frt_specgram {
    int fft_size = 0, hop = 0, nb = 0;
    double needed = 0.0;                         // fft_size * (1 - overlap), a float in the reference
    long long ring_len = 0, offset = 0, old_index = 0;
    frt_stft* stft = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DeviceBuffer ring, chunk, norm, jidx, dx, den, old_a, old_b, src, a, lut, pixels;
    bool old_is_a = true;
    int height = 0;
    bool has_map = false, has_lut = false;
    // Online_Linear_2D_resampler's scalars (online_linear_2D_resampler.py:21-43)
    double interp_L = 1.0, decim_M = 1.0, ratio = 1.0, orig_index = 0.0, resampled_index = 0.0;
    void* pin = nullptr;
    size_t pin_bytes = 0;
    void* pin_in = nullptr;
    size_t pin_in_bytes = 0;
    hipEvent_t in_done = nullptr;                // the last chunk has left the staging buffers (pin_in, chunk)
    bool in_pending = false;
    std::vector<int> h_src;
    std::vector<double> h_a;
};

extern "C" void frt_specgram_destroy(frt_specgram* h) {
    if (!h) return;
    if (h->stft) frt_stft_destroy(h->stft);
    DeviceBuffer* bufs[] = {&h->ring, &h->chunk, &h->norm, &h->jidx, &h->dx, &h->den, &h->old_a, &h->old_b, &h->src, &h->a, &h->lut,
                            &h->pixels};
    for (auto* b : bufs) b->release();
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->in_done) (void)hipEventDestroy(h->in_done);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int frt_specgram_create(frt_specgram** out, int fft_size, double overlap, int ring_length) {
    FRT_REQUIRE(out, "frt_specgram_create: null handle pointer");
    *out = nullptr;
    FRT_REQUIRE(overlap >= 0.0 && overlap < 1.0, "frt_specgram_create: overlap %g not in [0, 1)", overlap);
    frt_specgram* h = new frt_specgram();
    h->fft_size = fft_size;
    h->nb = fft_size / 2 + 1;
    h->needed = fft_size * (1.0 - overlap);                       // spectrogram.py:145
    h->hop = (int)h->needed;                                      // spectrogram.py:159: old_index += int(needed)
    h->ring_len = ring_length;
    int rc = FRT_OK;
    if (h->hop < 1 || ring_length < 2 * fft_size) {
        set_last_error("frt_specgram_create: hop %d < 1 or ring of %d samples shorter than two frames", h->hop, ring_length);
        rc = FRT_ERR_INVALID;
    }
    if (!rc) rc = frt_stft_create(&h->stft, fft_size, h->hop, 1, 64);
    if (!rc && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = FRT_ERR_HIP;
    if (!rc) {
        h->own_stream = true;
        rc = frt_stft_set_stream(h->stft, h->stream);
    }
    if (!rc && hipEventCreateWithFlags(&h->in_done, hipEventDisableTiming) != hipSuccess) rc = FRT_ERR_HIP;
    if (!rc) rc = h->ring.reserve(2 * (size_t)ring_length * sizeof(double));
    if (!rc && hipMemsetAsync(h->ring.ptr, 0, h->ring.bytes, h->stream) != hipSuccess) rc = FRT_ERR_HIP;
    if (rc) {
        frt_specgram_destroy(h);
        return rc;
    }
    *out = h;
    return FRT_OK;
}

extern "C" int frt_specgram_set_epilogue(frt_specgram* h, const double* weight_db, double spec_min, double spec_max, const uint32_t* lut256) {
    FRT_REQUIRE(h && lut256, "frt_specgram_set_epilogue: null argument");
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    int rc = frt_stft_set_epilogue(h->stft, weight_db, spec_min, spec_max, nullptr);
    if (rc) return rc;
    std::vector<uint32_t> l(lut256, lut256 + 256);
    if ((rc = upload(h->lut, l))) return rc;
    h->has_lut = true;
    return FRT_OK;
}

// Frequency map and screen height (Frequency_Resampler.setfreq / setnsamples + Online_Linear_2D_resampler.set_height): `freq`
// are the bin frequencies, `targets` the `height` frequencies of the screen rows.  A new height Fourier-resamples the
// carried column and restarts the time resampler's indices (online_linear_2D_resampler.py:45-55).
extern "C" int frt_specgram_set_screen(frt_specgram* h, const double* freq, const double* targets, int height) {
    FRT_REQUIRE(h && freq && targets && height >= 1, "frt_specgram_set_screen: bad arguments");
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    const int nb = h->nb;
    std::vector<int> j(height);
    std::vector<double> dx(height, 0.0), den(height, 1.0);
    interval_search(freq, nb, targets, height, j.data(), dx.data(), den.data());
    int rc;
    if ((rc = upload(h->jidx, j)) || (rc = upload(h->dx, dx)) || (rc = upload(h->den, den))) return rc;
    DeviceBuffer& cur = h->old_is_a ? h->old_a : h->old_b;
    DeviceBuffer& other = h->old_is_a ? h->old_b : h->old_a;
    if (h->height == 0) {
        std::vector<double> z(height, 0.0);                       // old_data = zeros(height)
        if ((rc = upload(cur, z)) || (rc = other.reserve((size_t)height * sizeof(double)))) return rc;
    } else if (height != h->height) {
        if ((rc = other.reserve((size_t)height * sizeof(double)))) return rc;
        if (height >= 2 && h->height >= 2) {
            if ((rc = frt_fourier_resample(cur.as<double>(), h->height, 1, other.as<double>(), height))) return rc;
        } else {
            FRT_HIP_CHECK(hipMemset(other.ptr, 0, (size_t)height * sizeof(double)));
        }
        h->old_is_a = !h->old_is_a;
        if ((rc = (h->old_is_a ? h->old_b : h->old_a).reserve((size_t)height * sizeof(double)))) return rc;
        h->orig_index = 0.0;
        h->resampled_index = 0.0;
    }
    h->height = height;
    h->has_map = true;
    return FRT_OK;
}

// Online_Linear_2D_resampler.set_ratio (online_linear_2D_resampler.py:35-43)
extern "C" int frt_specgram_set_ratio(frt_specgram* h, double interp_factor_L, double decim_factor_M) {
    FRT_REQUIRE(h && decim_factor_M != 0.0, "frt_specgram_set_ratio: bad arguments");
    if (h->interp_L != interp_factor_L || h->decim_M != decim_factor_M) {
        h->interp_L = interp_factor_L;
        h->decim_M = decim_factor_M;
        h->ratio = interp_factor_L / decim_factor_M;
        h->orig_index = 0.0;
        h->resampled_index = 0.0;
    }
    return FRT_OK;
}

// The ring grows like the reference's (x1.5 until the request fits, ringbuffer.py:102-130).
static int grow_ring(frt_specgram* h, long long need) {
    long long new_len = h->ring_len;
    while (new_len < need) new_len = (long long)(new_len * 1.5) + 1;
    DeviceBuffer grown;
    int rc;
    if ((rc = grown.reserve(2 * (size_t)new_len * sizeof(double)))) return rc;
    FRT_HIP_CHECK(hipMemsetAsync(grown.ptr, 0, grown.bytes, h->stream));
    const long long keep = h->offset < h->ring_len ? h->offset : h->ring_len;
    if (keep > 0)
        hipLaunchKernelGGL(ring_relay_kernel, dim3((unsigned)((keep + 255) / 256)), dim3(256), 0, h->stream, h->ring.as<double>(), h->ring_len,
                           grown.as<double>(), new_len, h->offset, keep);
    FRT_HIP_CHECK(hipGetLastError());
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->ring.release();
    h->ring = grown;
    grown.ptr = nullptr;
    grown.bytes = 0;
    h->ring_len = new_len;
    return FRT_OK;
}

// One chunk of new samples (host memory, float64).  pixels_out: [height][max_cols] uint32, rows top = highest frequency;
// *n_cols_out pixel columns were produced (0 is common), *n_frames_out spectra computed.
// Everything that can fail on the arguments is decided on copies of the scalar state BEFORE anything is changed: an
// error leaves the object as it was.
extern "C" int frt_specgram_push(frt_specgram* h, const double* chunk, int n, uint32_t* pixels_out, int max_cols, int* n_cols_out,
                                 int* n_frames_out) {
    FRT_REQUIRE(h && n >= 0 && n_cols_out, "frt_specgram_push: bad arguments");
    FRT_REQUIRE(h->has_map && h->has_lut, "frt_specgram_push: set_epilogue and set_screen first");
    FRT_REQUIRE(n == 0 || chunk, "frt_specgram_push: null chunk");
    *n_cols_out = 0;
    if (n_frames_out) *n_frames_out = 0;
    int rc;
    // ---- what this chunk will yield (spectrogram.py:133-159, online_linear_2D_resampler.py:61-97), on copies ----------
    const long long new_offset = h->offset + n;
    long long new_old_index = h->old_index;
    long long available = new_offset - new_old_index;
    if (available < 0) {
        available = 0;
        new_old_index = new_offset;
    }
    const int realizable = (int)std::floor((double)available / h->needed);
    double orig_index = h->orig_index, resampled_index = h->resampled_index;
    h->h_src.clear();
    h->h_a.clear();
    for (int jf = 0; jf < realizable; ++jf) {
        orig_index += 1.0;
        const int cnt = (int)std::ceil((orig_index - (resampled_index + h->ratio)) / h->ratio);      // processable(0)
        if (cnt <= 0) continue;
        double last_index = resampled_index;
        for (int k = 1; k <= cnt; ++k) {
            last_index = resampled_index + h->ratio * (double)k;
            h->h_a.push_back(orig_index - last_index);
            h->h_src.push_back(jf);
        }
        resampled_index = last_index;
    }
    const int n_out = (int)h->h_src.size();
    FRT_REQUIRE(n_out <= max_cols || !pixels_out, "frt_specgram_push: %d pixel columns, room for %d", n_out, max_cols);
    // the ring must hold the chunk and every frame it completes
    const long long span = realizable > 0 ? h->fft_size + (long long)(realizable - 1) * h->hop : 0;
    const long long last = new_old_index + (long long)(realizable - 1) * h->hop;
    long long need_len = (long long)n + h->fft_size;
    if (realizable > 0 && new_offset - (last - span) > need_len) need_len = new_offset - (last - span);
    if (need_len > h->ring_len && (rc = grow_ring(h, need_len))) return rc;

    if (n > 0) {
        // the staging buffers are reused from push to push: the previous chunk must have left them (a push that completes
        // no frame returns without waiting for its copy)
        if (h->in_pending) {
            FRT_HIP_CHECK(hipEventSynchronize(h->in_done));
            h->in_pending = false;
        }
        if ((rc = h->chunk.reserve((size_t)n * sizeof(double)))) return rc;
        // through pinned memory: a copy from pageable memory is staged by the runtime and blocks the calling thread
        if ((size_t)n * sizeof(double) > h->pin_in_bytes) {
            if (h->pin_in) (void)hipHostFree(h->pin_in);
            h->pin_in = nullptr;
            h->pin_in_bytes = 0;
            FRT_HIP_CHECK(hipHostMalloc(&h->pin_in, (size_t)n * sizeof(double) * 2, hipHostMallocDefault));
            h->pin_in_bytes = (size_t)n * sizeof(double) * 2;
        }
        memcpy(h->pin_in, chunk, (size_t)n * sizeof(double));
        // a widget-sized chunk is read by the ring write where it lies (page-locked memory is device accessible): one launch
        // instead of a copy and a launch; long chunks go up by the copy engine first
        const double* src = (const double*)h->pin_in;
        if ((size_t)n * sizeof(double) > kZeroCopyMax) {
            FRT_HIP_CHECK(hipMemcpyAsync(h->chunk.ptr, h->pin_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
            src = h->chunk.as<double>();
        }
        hipLaunchKernelGGL(ring_write_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, n, h->ring.as<double>(), h->ring_len,
                           h->offset);
        FRT_HIP_CHECK(hipGetLastError());
        if (realizable <= 0) {                     // a push that completes frames waits for the stream at its end anyway
            FRT_HIP_CHECK(hipEventRecord(h->in_done, h->stream));
            h->in_pending = true;
        }
    }
    h->offset = new_offset;
    h->old_index = new_old_index;
    if (realizable <= 0) return FRT_OK;
    // frame i holds the fft_size samples ending at old_index + i hop (data_indexed(old_index, fft_size), then old_index += hop)
    FRT_REQUIRE(span <= h->ring_len && h->offset - (last - span) <= h->ring_len, "frt_specgram_push: the ring no longer holds the frames");
    const long long stop0 = last % h->ring_len + h->ring_len;     // ringbuffer.py:91-92
    const double* window = h->ring.as<double>() + (stop0 - span);
    h->old_index += (long long)realizable * h->hop;
    if ((rc = h->norm.reserve((size_t)realizable * h->nb * sizeof(double)))) return rc;
    int64_t F = 0;
    if ((rc = frt_stft_run(h->stft, FRT_STFT_NORM, window, span, span, h->norm.ptr, &F))) return rc;
    FRT_REQUIRE(F == realizable, "frt_specgram_push: internal frame count %lld != %d", (long long)F, realizable);
    if (n_frames_out) *n_frames_out = realizable;

    h->orig_index = orig_index;                // the time resampler's scalars, as simulated above
    h->resampled_index = resampled_index;
    DeviceBuffer& old_in = h->old_is_a ? h->old_a : h->old_b;
    DeviceBuffer& old_out = h->old_is_a ? h->old_b : h->old_a;
    const int cols_alloc = n_out > 0 ? n_out : 1;
    if ((rc = h->src.reserve((size_t)cols_alloc * sizeof(int))) || (rc = h->a.reserve((size_t)cols_alloc * sizeof(double))) ||
        (rc = h->pixels.reserve((size_t)cols_alloc * h->height * sizeof(uint32_t))))
        return rc;
    ColumnTable inl{};
    const bool inline_cols = n_out <= kInlineCols;
    if (inline_cols) {
        for (int p = 0; p < n_out; ++p) {
            inl.src[p] = h->h_src[p];
            inl.a[p] = h->h_a[p];
        }
    } else {
        FRT_HIP_CHECK(hipMemcpyAsync(h->src.ptr, h->h_src.data(), (size_t)n_out * sizeof(int), hipMemcpyHostToDevice, h->stream));
        FRT_HIP_CHECK(hipMemcpyAsync(h->a.ptr, h->h_a.data(), (size_t)n_out * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    // pixels for a host caller: a block of a few columns is written by the kernel straight into the page-locked block
    const size_t pix_bytes = (size_t)n_out * h->height * 4;
    const bool host_out = n_out > 0 && pixels_out && !is_device_pointer(pixels_out);
    const bool direct_out = host_out && pix_bytes <= kZeroCopyMax;
    if (host_out && pix_bytes > h->pin_bytes) {
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->pin) (void)hipHostFree(h->pin);
        h->pin = nullptr;
        h->pin_bytes = 0;
        FRT_HIP_CHECK(hipHostMalloc(&h->pin, pix_bytes * 2, hipHostMallocDefault));
        h->pin_bytes = pix_bytes * 2;
    }
    hipLaunchKernelGGL(screen_columns_kernel, dim3((cols_alloc + 63) / 64, h->height), dim3(64), 0, h->stream, h->norm.as<double>(), h->nb,
                       realizable, h->jidx.as<int>(), h->dx.as<double>(), h->den.as<double>(), h->height, old_in.as<double>(),
                       old_out.as<double>(), inline_cols ? nullptr : h->src.as<int>(), inline_cols ? nullptr : h->a.as<double>(), inl, n_out,
                       h->lut.as<uint32_t>(), direct_out ? (uint32_t*)h->pin : h->pixels.as<uint32_t>(), n_out > 0 ? n_out : 1, 1);
    FRT_HIP_CHECK(hipGetLastError());
    h->old_is_a = !h->old_is_a;
    if (n_out > 0 && pixels_out) {
        if (is_device_pointer(pixels_out)) {
            FRT_HIP_CHECK(hipMemcpy2DAsync(pixels_out, (size_t)max_cols * 4, h->pixels.ptr, (size_t)n_out * 4, (size_t)n_out * 4, h->height,
                                           hipMemcpyDeviceToDevice, h->stream));
        } else {
            if (!direct_out) FRT_HIP_CHECK(hipMemcpyAsync(h->pin, h->pixels.ptr, pix_bytes, hipMemcpyDeviceToHost, h->stream));
            FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
            for (int r = 0; r < h->height; ++r)
                memcpy(pixels_out + (size_t)r * max_cols, (const uint32_t*)h->pin + (size_t)r * n_out, (size_t)n_out * 4);
        }
    }
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->in_pending = false;
    *n_cols_out = n_out;
    return FRT_OK;
}

extern "C" int frt_specgram_reset(frt_specgram* h) {
    FRT_REQUIRE(h, "frt_specgram_reset: null handle");
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    FRT_HIP_CHECK(hipMemset(h->ring.ptr, 0, h->ring.bytes));
    h->offset = h->old_index = 0;
    h->orig_index = h->resampled_index = 0.0;
    if (h->height) FRT_HIP_CHECK(hipMemset((h->old_is_a ? h->old_a : h->old_b).ptr, 0, (size_t)h->height * sizeof(double)));
    return FRT_OK;
}
