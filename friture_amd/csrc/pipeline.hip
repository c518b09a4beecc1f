// pipeline.hip — K6 and K4': the screen-space stages of the spectrogram and the block-wise
// exponential smoothing, for gfx950.  float64 like the reference; built with -ffp-contract=off so
// that every element is produced by the reference's IEEE operations in the reference's order
// (these stages end in an integer colour index, where a last-bit difference can flip a pixel).
//
// Reference semantics:
//   P5  Frequency_Resampler.push: per column np.interp(targets, freq, column)
//                                                     friture/signal/frequency_resampler.py:67-83
//   P6  Online_Linear_2D_resampler.push / linear_interp_2D: out = data (1 - a) + old a per emitted
//       pixel column        friture/signal/online_linear_2D_resampler.py:61-97, linear_interp.py:57-60
//   P7  Color_Transform.push: lut[int(clip(v, 0, 1) * 255)]
//                     friture/signal/color_tranform.py:48-51, friture/signal/lookup_table.py:50-52
//   P8  exp_smoothed_value(_2d): alpha * (data[:, :Nt] @ kernel[Nk-Nt:]) + previous * (1-alpha)^Nt
//                                                     friture/signal/exp_smoothing.py:40-56,91-107
#include <cmath>

#include "common.h"

namespace frt {

// np.interp with the interval index j[h] found on the host (the abscissae are plan constants):
//   j = -1 -> fp[0];  j >= n-1 -> fp[n-1];  xp[j] == x -> fp[j];
//   else slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j]);  slope * (x - xp[j]) + fp[j]
__global__ void freq_resample_kernel(const double* __restrict__ data, int n_bins, int n_cols, const int* __restrict__ jidx,
                                     const double* __restrict__ dx, const double* __restrict__ den, int height,
                                     double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (c >= n_cols) return;
    const int j = jidx[h];
    double v;
    if (j < 0) v = data[c];
    else if (j >= n_bins - 1) v = data[(size_t)(n_bins - 1) * n_cols + c];
    else {
        const double f0 = data[(size_t)j * n_cols + c];
        if (dx[h] == 0.0) v = f0;
        else {
            const double f1 = data[(size_t)(j + 1) * n_cols + c];
            const double slope = (f1 - f0) / den[h];
            v = slope * dx[h] + f0;
        }
    }
    out[(size_t)h * n_cols + c] = v;
}

__global__ void time_resample_kernel(const double* __restrict__ data, const double* __restrict__ old, int height, int n_cols,
                                     const int* __restrict__ src, const double* __restrict__ a, int n_out,
                                     double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (p >= n_out) return;
    const int c = src[p];
    const double cur = data[(size_t)h * n_cols + c];
    const double prev = c == 0 ? old[h] : data[(size_t)h * n_cols + c - 1];
    const double w = a[p];
    out[(size_t)h * n_out + p] = cur * (1.0 - w) + prev * w;
}

__global__ void colour_map_kernel(const uint32_t* __restrict__ lut, const double* __restrict__ v, long long count,
                                  uint32_t* __restrict__ out) {
    __shared__ uint32_t l[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) l[t] = lut[t];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        double x = v[i];
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);              // numpy.clip
        out[i] = l[(int)(x * 255.0)];
    }
}

// one wavefront per row: alpha * sum_t data[r][t] * kernel[off + t] + previous[r] * decay
__global__ void __launch_bounds__(64) exp_smooth_kernel(const double* __restrict__ data, long long row_stride, int nt,
                                                        const double* __restrict__ kern, double alpha, double decay,
                                                        const double* __restrict__ previous, double* __restrict__ out, int nf) {
    const int r = blockIdx.x;
    if (r >= nf) return;
    const double* row = data + (size_t)r * row_stride;
    double acc = 0.0;
    for (int t = threadIdx.x; t < nt; t += 64) acc += row[t] * kern[t];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (threadIdx.x == 0) out[r] = alpha * acc + previous[r] * decay;
}

// Ragged form: row r has its own length, taps, alpha — the octave-spectrum widget's bands (octavespectrum.py:103-112: one
// exp_smoothed_value per band, bands of an octave sharing kernel, alpha and length).  desc[r] = {data offset, taps offset, n,
// square}; the same products and the same summation order per row as exp_smooth_kernel, so a row equals its own call bit for bit.
struct ExpSmoothRow {
    long long data_off, kern_off;      // in doubles, into the call's packed data / taps
    int n, square;                     // samples used; 1: the datum is squared first (the widget smooths y^2)
    double alpha, decay;               // decay = (1 - alpha)^n, 0 when the row is longer than its kernel
};
__global__ void __launch_bounds__(64) exp_smooth_rows_kernel(const double* __restrict__ data, const double* __restrict__ taps,
                                                             const ExpSmoothRow* __restrict__ desc, const double* __restrict__ previous,
                                                             double* __restrict__ out, int rows) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const ExpSmoothRow d = desc[r];
    const double* row = data + d.data_off;
    const double* kern = taps + d.kern_off;
    double acc = 0.0;
    if (d.square) {
        for (int t = threadIdx.x; t < d.n; t += 64) {
            const double v = row[t];
            acc += (v * v) * kern[t];
        }
    } else {
        for (int t = threadIdx.x; t < d.n; t += 64) acc += row[t] * kern[t];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (threadIdx.x == 0) out[r] = d.n == 0 ? previous[r] : d.alpha * acc + previous[r] * d.decay;
}

// ---- Fourier resampling of a column (scipy_resample.py:51-141 as used by Online_Linear_2D_resampler.set_height,
// online_linear_2D_resampler.py:45-55): X = fft(x); keep the N = min(n, m) lowest frequencies; y = ifft(Y) * m / n.
// A resize event, n and m a few hundred to a few thousand and of any factorisation (screen heights): two direct DFT
// sums with table twiddles — thread k forms X[k] for the kept bins, thread j forms y[j] — O(n m) multiply-adds, no
// length restrictions, float64.
// kept bins in their order in Y: pos = (N + 1) / 2 from the front of X, neg = N - pos... the reference's slices are
//   Y[0 : (N+1)//2] = X[0 : (N+1)//2]   and   Y[-(N-1)//2 :] = X[-(N-1)//2 :]   with Python's floor division of the
// NEGATIVE number, i.e. the last ceil((N-1)/2) entries (for even N that includes the bin at -N/2).
__global__ void __launch_bounds__(256) fourier_bins_kernel(const double* __restrict__ x, int n, int count, int npos, int nneg,
                                                           const double* __restrict__ wn /* [n] cos, -sin of 2 pi r / n */,
                                                           double* __restrict__ X /* [count][npos + nneg] complex */) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (b >= npos + nneg) return;
    const int k = b < npos ? b : n - nneg + (b - npos);          // bin of X
    const double* xv = x + (size_t)v * n;
    double re = 0.0, im = 0.0;
    int r = 0;                                                   // (k t) mod n
    for (int t = 0; t < n; ++t) {
        re += xv[t] * wn[2 * r];
        im += xv[t] * wn[2 * r + 1];
        r += k;
        if (r >= n) r -= n;
    }
    double* o = X + ((size_t)v * (npos + nneg) + b) * 2;
    o[0] = re;
    o[1] = im;
}

__global__ void __launch_bounds__(256) fourier_synth_kernel(const double* __restrict__ X, int n, int m, int count, int npos, int nneg,
                                                            const double* __restrict__ wm /* [m] cos, +sin of 2 pi r / m */,
                                                            double* __restrict__ y /* [count][m] */) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (j >= m) return;
    const double* Xv = X + (size_t)v * (npos + nneg) * 2;
    double acc = 0.0;
    for (int b = 0; b < npos + nneg; ++b) {
        const int kk = b < npos ? b : m - nneg + (b - npos);     // position in Y
        const int r = (int)(((long long)kk * j) % m);
        acc += Xv[2 * b] * wm[2 * r] - Xv[2 * b + 1] * wm[2 * r + 1];      // Re(X e^{+i...})
    }
    // ifft's 1/m times the reference's m/n
    y[(size_t)v * m + j] = acc * ((1.0 / (double)m) * ((double)m / (double)n));
}

}  // namespace frt

using namespace frt;

extern "C" int frt_freq_resample(const double* freq, int n_bins, const double* targets, int height, const double* data,
                                 int n_cols, double* out) {
    FRT_REQUIRE(freq && targets && n_bins >= 1 && height >= 1 && n_cols >= 0, "frt_freq_resample: bad arguments");
    if (n_cols == 0) return FRT_OK;
    FRT_REQUIRE(data && out, "frt_freq_resample: null buffer");
    FRT_REQUIRE(!is_device_pointer(freq) && !is_device_pointer(targets), "frt_freq_resample: freq/targets are host tables");
    // interval search on the host: numpy's binary search semantics (largest j with freq[j] <= x)
    std::vector<int> j(height);
    std::vector<double> dx(height, 0.0), den(height, 1.0);
    for (int h = 0; h < height; ++h) {
        const double x = targets[h];
        if (!(x >= freq[0])) { j[h] = -1; continue; }                       // left of the table (or NaN)
        if (x > freq[n_bins - 1]) { j[h] = n_bins; continue; }
        int lo = 0, hi = n_bins;                                             // freq[lo] <= x < freq[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) / 2;
            if (freq[mid] <= x) lo = mid; else hi = mid;
        }
        j[h] = lo;
        if (lo < n_bins - 1) {
            dx[h] = x - freq[lo];
            den[h] = freq[lo + 1] - freq[lo];
        }
    }
    StageCall st;
    const int i_data = st.add_in(data, (size_t)n_bins * n_cols * sizeof(double)), i_j = st.add_in(j.data(), (size_t)height * sizeof(int)),
              i_dx = st.add_in(dx.data(), (size_t)height * sizeof(double)), i_den = st.add_in(den.data(), (size_t)height * sizeof(double)),
              i_out = st.add_out(out, (size_t)height * n_cols * sizeof(double));
    int rc;
    if ((rc = st.begin())) return rc;
    hipLaunchKernelGGL(freq_resample_kernel, dim3((n_cols + 63) / 64, height), dim3(64), 0, st.stream(), st.ptr<const double>(i_data), n_bins,
                       n_cols, st.ptr<const int>(i_j), st.ptr<const double>(i_dx), st.ptr<const double>(i_den), height, st.ptr<double>(i_out));
    return st.finish();
}

extern "C" int frt_time_resample(const double* data, const double* old, int height, int n_cols, const int* src_col,
                                 const double* a, int n_out, double* out) {
    FRT_REQUIRE(height >= 1 && n_cols >= 0 && n_out >= 0, "frt_time_resample: bad arguments");
    if (n_out == 0) return FRT_OK;
    FRT_REQUIRE(data && old && src_col && a && out, "frt_time_resample: null buffer");
    FRT_REQUIRE(!is_device_pointer(src_col) && !is_device_pointer(a), "frt_time_resample: src_col/a are host tables");
    for (int p = 0; p < n_out; ++p) FRT_REQUIRE(src_col[p] >= 0 && src_col[p] < n_cols, "frt_time_resample: source column out of range");
    StageCall st;
    const int i_data = st.add_in(data, (size_t)height * n_cols * sizeof(double)), i_old = st.add_in(old, (size_t)height * sizeof(double)),
              i_src = st.add_in(src_col, (size_t)n_out * sizeof(int)), i_a = st.add_in(a, (size_t)n_out * sizeof(double)),
              i_out = st.add_out(out, (size_t)height * n_out * sizeof(double));
    int rc;
    if ((rc = st.begin())) return rc;
    hipLaunchKernelGGL(time_resample_kernel, dim3((n_out + 63) / 64, height), dim3(64), 0, st.stream(), st.ptr<const double>(i_data),
                       st.ptr<const double>(i_old), height, n_cols, st.ptr<const int>(i_src), st.ptr<const double>(i_a), n_out,
                       st.ptr<double>(i_out));
    return st.finish();
}

extern "C" int frt_colour_map(const uint32_t* lut256, const double* values, int64_t count, uint32_t* out) {
    FRT_REQUIRE(lut256 && count >= 0, "frt_colour_map: bad arguments");
    if (count == 0) return FRT_OK;
    FRT_REQUIRE(values && out, "frt_colour_map: null buffer");
    StageCall st;
    const int i_lut = st.add_in(lut256, 256 * sizeof(uint32_t)), i_v = st.add_in(values, (size_t)count * sizeof(double)),
              i_out = st.add_out(out, (size_t)count * sizeof(uint32_t));
    int rc;
    if ((rc = st.begin())) return rc;
    long long blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(colour_map_kernel, dim3((unsigned)blocks), dim3(256), 0, st.stream(), st.ptr<const uint32_t>(i_lut),
                       st.ptr<const double>(i_v), (long long)count, st.ptr<uint32_t>(i_out));
    return st.finish();
}

extern "C" int frt_exp_smooth_2d(const double* kernel, int nk, double alpha, const double* data, int nf, int nt, int64_t row_stride,
                                 const double* previous, double* out) {
    FRT_REQUIRE(kernel && nk >= 0 && nf >= 0 && nt >= 0 && row_stride >= nt, "frt_exp_smooth_2d: bad arguments");
    if (nf == 0) return FRT_OK;
    FRT_REQUIRE(previous && out && (nt == 0 || data), "frt_exp_smooth_2d: null buffer");
    FRT_REQUIRE(!is_device_pointer(kernel), "frt_exp_smooth_2d: kernel is a host table");
    // exp_smoothing.py:94-101: more data than kernel taps -> only the first Nk samples count and
    // the previous value is forgotten
    int n = nt;
    double decay;
    if (n > nk) {
        n = nk;
        decay = 0.0;
    } else {
        decay = std::pow(1.0 - alpha, (double)n);
    }
    if (n == 0) {        // exp_smoothing.py:103-104: nothing new, return a copy of previous
        FRT_HIP_CHECK(hipMemcpy(out, previous, (size_t)nf * sizeof(double), hipMemcpyDefault));
        return FRT_OK;
    }
    StageCall st;
    const int i_k = st.add_in(kernel + (nk - n), (size_t)n * sizeof(double)), i_prev = st.add_in(previous, (size_t)nf * sizeof(double)),
              i_data = st.add_in(data, ((size_t)(nf - 1) * row_stride + nt) * sizeof(double)), i_out = st.add_out(out, (size_t)nf * sizeof(double));
    int rc;
    if ((rc = st.begin())) return rc;
    hipLaunchKernelGGL(exp_smooth_kernel, dim3(nf), dim3(64), 0, st.stream(), st.ptr<const double>(i_data), (long long)row_stride, n,
                       st.ptr<const double>(i_k), alpha, decay, st.ptr<const double>(i_prev), st.ptr<double>(i_out), nf);
    return st.finish();
}

extern "C" int frt_exp_smooth_groups(int n_groups, const double* const* kernels, const int* nk, const double* alphas,
                                     const double* const* data, const int* nf, const int* nt, const int64_t* row_stride, int square,
                                     const double* previous, double* out) {
    FRT_REQUIRE(n_groups >= 0 && n_groups <= 64, "frt_exp_smooth_groups: %d groups (at most 64)", n_groups);
    if (n_groups == 0) return FRT_OK;
    FRT_REQUIRE(kernels && nk && alphas && data && nf && nt && row_stride && previous && out, "frt_exp_smooth_groups: null argument");
    FRT_REQUIRE(!is_device_pointer(previous) && !is_device_pointer(out), "frt_exp_smooth_groups: host arrays (the widget's chunk)");
    int rows = 0;
    for (int g = 0; g < n_groups; ++g) {
        FRT_REQUIRE(nk[g] >= 0 && nf[g] >= 0 && nt[g] >= 0 && row_stride[g] >= nt[g], "frt_exp_smooth_groups: bad shape in group %d", g);
        FRT_REQUIRE(kernels[g] && (nt[g] == 0 || nf[g] == 0 || data[g]), "frt_exp_smooth_groups: null buffer in group %d", g);
        FRT_REQUIRE(!is_device_pointer(kernels[g]) && (nt[g] == 0 || nf[g] == 0 || !is_device_pointer(data[g])),
                    "frt_exp_smooth_groups: host arrays (the widget's chunk)");
        rows += nf[g];
    }
    if (rows == 0) return FRT_OK;
    StageCall st;
    std::vector<ExpSmoothRow> desc((size_t)rows);
    std::vector<int> in_data(n_groups, -1), in_taps(n_groups, -1);
    std::vector<int> used(n_groups);
    for (int g = 0; g < n_groups; ++g) {
        // exp_smoothing.py:94-101: more data than taps -> only the first Nk samples count and the previous value is forgotten
        used[g] = nt[g] > nk[g] ? nk[g] : nt[g];
        if (nf[g] == 0 || used[g] == 0) continue;
        in_taps[g] = st.add_in(kernels[g] + (nk[g] - used[g]), (size_t)used[g] * sizeof(double));
        in_data[g] = st.add_in(data[g], ((size_t)(nf[g] - 1) * row_stride[g] + nt[g]) * sizeof(double));
    }
    const int i_prev = st.add_in(previous, (size_t)rows * sizeof(double));
    const int i_desc = st.add_in(desc.data(), desc.size() * sizeof(ExpSmoothRow));      // (filled below, before begin() copies it)
    const int i_out = st.add_out(out, (size_t)rows * sizeof(double));
    // every staged input lies in one block in registration order at offsets fixed by add_in: the rows carry their distances from
    // the first one, the kernel gets its address
    int first = -1;
    for (int g = 0; g < n_groups && first < 0; ++g) first = in_taps[g];
    if (first < 0) {          // every row is empty: exp_smoothing.py:103-104, a copy of previous
        memcpy(out, previous, (size_t)rows * sizeof(double));
        return FRT_OK;
    }
    int r = 0;
    for (int g = 0; g < n_groups; ++g) {
        const double decay = nt[g] > nk[g] ? 0.0 : std::pow(1.0 - alphas[g], (double)used[g]);
        for (int f = 0; f < nf[g]; ++f, ++r) {
            ExpSmoothRow& d = desc[(size_t)r];
            d.n = used[g];
            d.square = square ? 1 : 0;
            d.alpha = alphas[g];
            d.decay = decay;
            d.kern_off = d.n ? (long long)((st.offset(in_taps[g]) - st.offset(first)) / sizeof(double)) : 0;
            d.data_off = d.n ? (long long)((st.offset(in_data[g]) - st.offset(first)) / sizeof(double)) + (long long)f * row_stride[g] : 0;
        }
    }
    int rc;
    if ((rc = st.begin())) return rc;
    const double* base = st.ptr<const double>(first);
    hipLaunchKernelGGL(exp_smooth_rows_kernel, dim3(rows), dim3(64), 0, st.stream(), base, base, st.ptr<const ExpSmoothRow>(i_desc),
                       st.ptr<const double>(i_prev), st.ptr<double>(i_out), rows);
    return st.finish();
}


extern "C" int frt_fourier_resample(const double* x, int n, int count, double* y, int m) {
    // (min(n, m) = 1 is degenerate in the reference: its slice -(N-1)//2 = 0 selects the whole array, which broadcasts for
    // n = 1 and raises for n > 1; the Python mirror reproduces both without a kernel)
    FRT_REQUIRE(n >= 2 && m >= 2 && count >= 0 && n <= (1 << 20) && m <= (1 << 20), "frt_fourier_resample: bad lengths %d -> %d", n, m);
    if (count == 0) return FRT_OK;
    FRT_REQUIRE(x && y, "frt_fourier_resample: null buffer");
    const int N = n < m ? n : m;
    const int npos = (N + 1) / 2, nneg = N / 2;                 // ceil((N - 1) / 2) = N / 2
    const long double pi2 = 6.283185307179586476925286766559L;
    std::vector<double> wn(2 * (size_t)n), wm(2 * (size_t)m);
    for (int r = 0; r < n; ++r) {
        wn[2 * r] = (double)cosl(pi2 * r / n);
        wn[2 * r + 1] = (double)(-sinl(pi2 * r / n));
    }
    for (int r = 0; r < m; ++r) {
        wm[2 * r] = (double)cosl(pi2 * r / m);
        wm[2 * r + 1] = (double)sinl(pi2 * r / m);
    }
    StageCall st;
    const int i_x = st.add_in(x, (size_t)count * n * sizeof(double)), i_wn = st.add_in(wn.data(), wn.size() * sizeof(double)),
              i_wm = st.add_in(wm.data(), wm.size() * sizeof(double)), i_X = st.add_scratch((size_t)count * N * 2 * sizeof(double)),
              i_y = st.add_out(y, (size_t)count * m * sizeof(double));
    int rc;
    if ((rc = st.begin())) return rc;
    hipLaunchKernelGGL(fourier_bins_kernel, dim3((N + 255) / 256, count), dim3(256), 0, st.stream(), st.ptr<const double>(i_x), n, count, npos,
                       nneg, st.ptr<const double>(i_wn), st.ptr<double>(i_X));
    hipLaunchKernelGGL(fourier_synth_kernel, dim3((m + 255) / 256, count), dim3(256), 0, st.stream(), st.ptr<const double>(i_X), n, m, count,
                       npos, nneg, st.ptr<const double>(i_wm), st.ptr<double>(i_y));
    return st.finish();
}
