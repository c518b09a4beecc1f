/* friture_hip.h — C ABI of libfriture_hip.so, the MI355X (gfx950) backend for Friture's
 * spectral-analysis hot path.
 *
 * The reference (tlecomte/friture) has no FFI: its boundary is a set of duck-typed Python
 * classes and functions.  Every entry point below names the reference interface
 * it replaces (paths relative to the reference checkout); the ctypes bindings that present the
 * reference's own names on top of this ABI live in friture_amd/ and are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success and a negative status on failure, never throws;
 *     frt_last_error() returns a thread-local description of the last failure;
 *   - handles are opaque and own all device memory; one handle per Python object, used from
 *     one thread at a time (the reference calls everything from the Qt GUI thread);
 *   - buffers are caller owned.  A buffer argument may be a host pointer (staged through an
 *     internal device buffer, call returns after the result is back on the host) or a device
 *     pointer (kernels are enqueued on the handle's stream and the call returns immediately);
 *     inputs and outputs of one call must live on the same side;
 *   - audio is channel-major: x[c][t].
 */
#ifndef FRITURE_HIP_H
#define FRITURE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define FRT_OK 0
#define FRT_ERR_INVALID (-1)
#define FRT_ERR_HIP (-2)
#define FRT_ERR_NO_DEVICE (-3)
#define FRT_ERR_UNSUPPORTED (-4)
#define FRT_ERR_TOO_SMALL (-5)

/* ---- library ---------------------------------------------------------------------------- */

/* Binds the calling process to HIP device `device` (one process per GPU) and verifies that it is
 * a gfx950 part.  n_cus_out / hbm_bytes_out may be NULL. */
int frt_init(int device, int* n_cus_out, int64_t* hbm_bytes_out);
/* The same two figures for any visible device WITHOUT binding the process to it (the current device is untouched). */
int frt_device_properties(int device, int* n_cus_out, int64_t* hbm_bytes_out);
const char* frt_last_error(void);
const char* frt_version(void);
/* 1 if `p` is device memory, 0 if host memory. */
int frt_is_device_pointer(const void* p);
/* Path selection for tests and A/B runs.  The library never reads the environment; where one entry point has two product
 * code paths chosen by the shape of the call, this forces one of them for the whole process (value < 0: back to the shape
 * rule).  Both paths of every option compute the same reference function and are held to the same parity bar:
 *   "gcc_one_workgroup"    frt_gcc_phat: 1 = one workgroup per window pair whatever the batch size, 0 = a pair as launches
 *                          of its phases (default: by batch size)
 *                          (also read by frt_gcc_create: a handle of at most CUs / 8 pairs of the default window made while the
 *                          option is 1 gets the two-way plan of the large batches instead of the four-way plan of the small ones)
 *   "gcc_resident"         frt_gcc_phat, default window (24000 samples), one workgroup per pair: 0 = the kernel that parks the
 *                          sub-spectra in a scratch slab in HBM instead of the one that keeps them in registers and LDS
 *   "ola_defer"            frt_octbank_filter / _energies (mode 1, batched, >= 6 bands per octave): 0 = every stage's band filters in the
 *                          stage's own launch instead of one deferred launch over the low-rate stages
 *   "iir_lookback"         frt_octbank_energies (mode 0, time-parallel): 0 = every stage's chunk start states from a scan launch
 *                          instead of the output pass's own look-back over its predecessors' end states at the high-rate stages
 *   "iir_lane_columns"     frt_octbank_energies (mode 0, time-parallel): the output pass with the filter groups of 64 chunks in one
 *                          workgroup and their samples staged through LDS — 1 = at every stage it serves, 0 = at none (default:
 *                          where the launch has a workgroup per compute unit and the groups share enough samples)
 *   "gcc_any_length"       frt_gcc_create: 1 = the chirp-z transform also for lengths the mixed-radix plan serves
 *   "ola_chunk_kernels"    frt_octbank_filter (mode 1, one block of <= 1024 host samples): 0 = per-stage transform launches
 *                          instead of the two running-convolution launches
 *   "pitch_grid_two_pass"  frt_pitch_track: 1 = the two-pass log-grid kernel also on the widget's 1023-point grid
 * Unknown names: FRT_ERR_INVALID. */
int frt_set_option(const char* name, int value);
int frt_get_option(const char* name, int* value_out);

/* ---- K1: STFT -> power spectrum ( -> dB / normalised / colour pixels) -------------------------
 * Replaces audioproc.analyzelive + norm_square (friture/audioproc.py:42-50) looped by the STFT
 * drivers Spectrum_Widget.handle_new_data (friture/spectrum.py:144-155) and
 * Spectrogram_Widget.handle_new_data (friture/spectrogram.py:149-159), and the dB / weighting /
 * normalise / colour-LUT steps that follow (friture/spectrogram.py:119-129,161-162,
 * friture/signal/color_tranform.py:48-51, friture/signal/lookup_table.py:50-52).
 *
 * Frame f of channel c covers x[c][f*hop .. f*hop + fft_size); n_frames = (T - fft_size)/hop + 1.
 * Outputs are frame-major: out[c][f][k], k = 0..fft_size/2 (the reference's (N/2+1, frames)
 * array is the transpose of one channel's slab). */
typedef struct frt_stft frt_stft;

/* output kinds for frt_stft_run */
#define FRT_STFT_PSD 0   /* float: |rfft(x*w)|^2 / N^2                      audioproc.py:44-50  */
#define FRT_STFT_DB 1    /* float: 10 log10(psd + 1e-30) + weight[k]        spectrum.py:95-101  */
#define FRT_STFT_NORM 2  /* float: (dB - spec_min)/(spec_max - spec_min)    spectrogram.py:128  */
#define FRT_STFT_IMAGE 3 /* uint32: lut[int(clip(norm,0,1)*255)]            lookup_table.py:50  */

/* fft_size: power of two in [32, 16384] (spectrum_settings.py:61-70); hop in [1, ...);
 * precision: 32 (float in, float arithmetic) or 64 (double in, double arithmetic, double out). */
int frt_stft_create(frt_stft** h, int fft_size, int hop, int n_channels, int precision);
void frt_stft_destroy(frt_stft* h);
/* hipStream_t used for device-pointer calls (NULL = default stream). */
int frt_stft_set_stream(frt_stft* h, void* hip_stream);
/* weight_db: fft_size/2+1 values added to the dB spectrum or NULL (no weighting);
 * lut256: 256 colour words or NULL. */
int frt_stft_set_epilogue(frt_stft* h, const double* weight_db, double spec_min, double spec_max,
                          const uint32_t* lut256);
/* x: [n_channels][x_stride] samples (float for precision 32, double for 64), T valid per channel.
 * out: [n_channels][n_frames][fft_size/2+1] elements of the output kind (4 bytes each at
 * precision 32; at precision 64 PSD/DB/NORM are doubles and IMAGE stays uint32).
 * x and out: both host (returns with the result), both device (enqueued on the handle's stream), or x host and out
 * device (the samples go up through page-locked memory, the call returns without waiting: order consumers on the stream). */
int frt_stft_run(frt_stft* h, int kind, const void* x, int64_t T, int64_t x_stride, void* out,
                 int64_t* n_frames_out);
/* The same transform with SPLIT output rows (fft_size <= 1024; FRT_ERR_UNSUPPORTED above): out_rows[c][f][k] holds
 * bins k = 0..fft_size/2-1 — rows of fft_size/2 values, so that every row is a whole number of 64-byte lines when
 * out_rows is 64-byte aligned — and out_nyquist[c][f] holds bin fft_size/2 of every frame (element types as for
 * frt_stft_run).  Values are bit-identical to frt_stft_run's; only their addresses differ.  The reference's own array is
 * (N/2+1, frames), frequency-major (friture/spectrogram.py:149-159, friture/spectrum.py:144-155), so neither layout is
 * its transpose-free image: the packed rows of frt_stft_run are what the drop-in classes hand on, the split rows are for
 * batch consumers — a row store of the packed layout starts on a 4-byte boundary and touches five 64-byte lines where
 * this one touches four.  Same host / device pointer rules as frt_stft_run; out_rows and out_nyquist
 * on the same side. */
int frt_stft_run_split(frt_stft* h, int kind, const void* x, int64_t T, int64_t x_stride, void* out_rows,
                       void* out_nyquist, int64_t* n_frames_out);
/* survey-named conveniences (precision-32 handles): */
int frt_stft_psd(frt_stft* h, const float* x, int64_t T, float* psd_out, int64_t* n_frames_out);
int frt_stft_image(frt_stft* h, const float* x, int64_t T, uint32_t* rgba_out, int64_t* n_frames_out);
/* Exact drop-in for audioproc.analyzelive (audioproc.py:42-47): one double frame of fft_size
 * samples -> fft_size/2+1 doubles.  Requires a precision-64 handle. */
int frt_stft_analyzelive_f64(frt_stft* h, const double* frame, double* psd_out);
/* Number of frames frt_stft_run produces for T samples per channel. */
int64_t frt_stft_frames_for(const frt_stft* h, int64_t T);
/* Tuning hook (bench / tests): frames a workgroup lane-group processes back to back; 0 = auto.  A negative
 * value sets the run length to its magnitude and, for fft_size >= 2048, selects the generic workgroup-wide
 * kernel instead of the radix-16 + wave-local one (A/B runs). */
int frt_stft_set_run_length(frt_stft* h, int frames_per_run);

/* ---- K2 / K4: octave filter bank with decimation, band energies -------------------------------
 * mode 0 replaces the exact IIR bank octave_filter_bank_decimation (friture/filter.py:86-118) built
 * on lfilter_float64_1D (friture/signal/lfilter.py:85-147) and decimate
 * (friture/signal/decimate.py:27-42): 9 octaves, `bands_per_octave` 4th-order band-passes per
 * octave on the j-times decimated signal, 12th-order elliptic decimator between octaves, carried
 * state.  Sequential mode is bit-identical to the reference; the time-parallel mode
 * (frt_octbank_set_chunk) evaluates the same recurrences chunk-wise and agrees to rounding.
 * mode 1 is the FFT overlap-add bank of Octave_Filters.filter (friture/octavefilters.py:49-58 /
 * friture/filter.py:136-247): 512-tap minimum-phase FIR of every IIR, 511-sample tails carried across calls; the same
 * frt_octbank_filter / frt_octbank_energies entry points.  Up to 1024 samples a call is one overlap-add block at the
 * reference's FFT sizes; a longer input is processed as if fed in 1024-sample blocks, every octave stage as one launch
 * over all blocks (the reference itself would crop such an input to its first stage's FFT size).
 *
 * Band order everywhere is the reference's list order: band k = 0 is the lowest band
 * (dec = 256), band 9*bpo-1 the highest (dec = 1); dec[k] = 2^(8 - k / bpo). */
typedef struct frt_octbank frt_octbank;

/* boct/aoct: [bands_per_octave][5]; bdec/adec: [13] (friture/generated_filters.py PARAMS);
 * boct_fir/bdec_fir are only used by mode 1 and may be NULL.  bands_per_octave = 0 creates a
 * decimator-only handle for frt_decimate_multiple. */
int frt_octbank_create(frt_octbank** h, int bands_per_octave, int n_channels, int mode, const double* boct,
                       const double* aoct, const double* bdec, const double* adec, const double* boct_fir,
                       const double* bdec_fir);
void frt_octbank_destroy(frt_octbank* h);
int frt_octbank_set_stream(frt_octbank* h, void* hip_stream);
/* zero every carried state (octave_filter_bank_decimation_filtic, friture/filter.py:121-133) */
int frt_octbank_reset(frt_octbank* h);
/* 0 (default): sequential in time, bit-identical to the reference.  chunk0 > 0 (multiple of 64, >= 256; for
 * frt_octbank_energies a multiple of the energy block, or a divisor of it — the latter for 16-byte aligned device
 * input in whole chunks only): batches of at least 2*chunk0 samples are
 * processed time-parallel, octave stage j in chunks of max(64, chunk0 / 2^j) of its own samples.  The two modes
 * agree to ~1e-10 of the input scale (same recurrences, other association order).  A negative value selects
 * the same chunking with the zero-state pass run as a second recurrence instead of a table product (A/B runs). */
int frt_octbank_set_chunk(frt_octbank* h, int chunk0);
/* doubles per channel in the packed output for n input samples: sum over bands of ceil(n / dec) */
int64_t frt_octbank_packed_length(const frt_octbank* h, int n);
/* x: [n_channels][n]; y_packed: [n_channels][packed_length], band k stored after band k-1;
 * dec_out: [9*bpo] or NULL.  n = 0 -> FRT_ERR_TOO_SMALL ("Filter input is too small",
 * friture/signal/decimate.py:33-34). */
int frt_octbank_filter(frt_octbank* h, const double* x, int n, double* y_packed, int* dec_out);
/* carried filter states per channel in the reference's zis/zfs order (friture/filter.py:121-133):
 * per octave the band states i = bpo-1..0 (4 doubles each) then the decimator's 12. */
int frt_octbank_state_length(const frt_octbank* h);
int frt_octbank_get_state(frt_octbank* h, double* z /* [n_channels][state_length] */);
int frt_octbank_set_state(frt_octbank* h, const double* z);
/* Band energies of OctaveSpectrum_Widget.handle_new_data (friture/octavespectrum.py:101-121) for a
 * whole batch: x float [n_channels][n] is cut in blocks of `block` samples (power of two >= 256, the
 * widget's chunk), per block and band sp = alpha*sum_i (1-alpha)^(m-1-i) y_i^2 + sp_prev*(1-alpha)^m
 * (friture/signal/exp_smoothing.py:40-56; m = block/dec), carried across blocks and calls.
 * energy_out: float [n_channels][n/block][9*bpo]; as_db != 0 stores 10 log10(sp + 1e-30) + weight_db[k]
 * (weight_db may be NULL) instead of sp.  The band signals themselves are not written.  Mode 1: block <= 1024. */
int frt_octbank_energies(frt_octbank* h, const float* x, int64_t n, int block, const double* alphas,
                         const double* weight_db, int as_db, float* energy_out);
/* decimate_multiple (friture/signal/decimate.py:45-71) with carried state: n_stages chained
 * decimations by 2 of x [n_channels][n] -> out [n_channels][*n_out].  Needs a bands_per_octave = 0 handle. */
int frt_decimate_multiple(frt_octbank* h, int n_stages, const double* x, int n, double* out, int* n_out);
/* the same with the reference's functional interface (the states are arguments and results, decimate.py:45-71) as ONE call
 * on HOST arrays: x [n_channels][n] -> out [n_channels][*n_out], zi / zf [n_channels][n_stages][order] (order = the handle's
 * decimator order: 12 for the bank's) or NULL (zero state / not wanted); the handle's carried state is not touched.  Every
 * channel of the handle rides in the same n_stages launches (the delay estimator's two channels, delay_estimator.py:97-98).
 * The samples are read and the result written in place in page-locked memory: roundup(8 n channels, 256) + 128 n_stages
 * channels bytes must fit 256 KB, i.e. n <= 32 639 for one channel and two stages (FRT_ERR_INVALID above that: use
 * frt_octbank_set_state + frt_decimate_multiple). */
int frt_decimate_multiple_state(frt_octbank* h, int n_stages, const double* x, int n, const double* zi, double* out, int* n_out,
                                double* zf);
/* lfilter_float64_1D (friture/signal/lfilter.py:85-147): direct form II transposed IIR of one host
 * signal with explicit state; len(b) = len(a) = n_coef <= 16, zi/zf hold n_coef-1 doubles. */
int frt_lfilter_f64(const double* b, const double* a, int n_coef, const double* x, int n, const double* zi,
                    double* y, double* zf);

/* ---- the delay estimator's per-chunk part, device resident (SURVEY.md §8 row 5 caller) -----------------------------
 * Delay_Estimator_Widget.handle_new_data (friture/delay_estimator.py:87-131) without its GCC-PHAT call: per chunk of both
 * channels, n_stages chained decimations by 2 with carried state (:97-98, decimate_multiple) and a push into two private
 * mirror rings of the decimated signals (:99-100, friture/ringbuffer.py:39-63), all in HBM.  frt_delay_push is
 * asynchronous (a pinned slot, the decimation stages and one ring-write launch on the object's stream; nothing comes back
 * per chunk).  Windows are handed out as device pointers into the rings (RingBuffer.data_indexed, ringbuffer.py:87-99; the
 * rings grow by the reference's rule, ringbuffer.py:102-130) and go to frt_gcc_phat / frt_gcc_readout as they are. */
typedef struct frt_delay frt_delay;
int frt_delay_create(frt_delay** out, const double* bdec, const double* adec, int n_stages, int ring_length /* 10000 */);
void frt_delay_destroy(frt_delay* h);
void* frt_delay_stream(frt_delay* h);                 /* the hipStream_t the pushes are enqueued on */
/* x: float64 [2][n] in HOST memory; *offset_out = decimated samples pushed so far (RingBuffer.offset), may be NULL */
int frt_delay_push(frt_delay* h, const double* x, int n, int64_t* offset_out);
/* the rings hold at least `length` samples from now on (RingBuffer.grow_if_needed) */
int frt_delay_reserve(frt_delay* h, int length);
/* device pointers of the `length` samples of each ring that end at absolute index `end` (<= offset); valid until the
 * rings grow.  Enqueued pushes may still be in flight: order consumers on frt_delay_stream or call frt_delay_window_std. */
int frt_delay_window(frt_delay* h, int64_t end, int length, double** d0, double** d1);
/* numpy.std of both windows (the gate of delay_estimator.py:127) -> std_out[2] on the host; waits for the object's stream */
int frt_delay_window_std(frt_delay* h, const double* d0, const double* d1, int length, double* std_out);
/* d0 -= means[0], d1 -= means[1] in place (generalized_cross_correlation does this to its views, correlation.py:27-28);
 * means: two doubles in DEVICE memory (frt_gcc_phat's means_out); stream: the stream frt_gcc_phat ran on, NULL = the object's */
int frt_delay_demean(frt_delay* h, double* d0, double* d1, int length, const double* means, void* stream);

/* ---- device-resident streaming spectrogram ------------------------------------------------
 * One object does what Spectrogram_Widget.handle_new_data does per chunk (friture/spectrogram.py:131-177) and hands back
 * the block CanvasScaledSpectrogram.addData draws (friture/spectrogram_image.py:82-129): a mirror ring of the samples in
 * HBM (friture/ringbuffer.py:39-63; only the new chunk crosses PCIe), the realizable frames through the float64 STFT with
 * the dB / weighting / normalise epilogue, np.interp onto the screen rows (friture/signal/frequency_resampler.py:67-83),
 * the online time resampler with its carried column on the device (friture/signal/online_linear_2D_resampler.py:45-97),
 * clip + LUT (friture/signal/color_tranform.py:48-51) and the flip of the frequency axis.  Float64, the reference's
 * operations in the reference's order: the pixels are the reference's pixels.
 * overlap as in the widget (0.75 default): needed = fft_size (1 - overlap), hop = int(needed).  ring_length >= 2 fft_size. */
typedef struct frt_specgram frt_specgram;
int frt_specgram_create(frt_specgram** h, int fft_size, double overlap, int ring_length);
void frt_specgram_destroy(frt_specgram* h);
/* weight_db [fft_size/2+1] or NULL, dB range, 256 colour words */
int frt_specgram_set_epilogue(frt_specgram* h, const double* weight_db, double spec_min, double spec_max, const uint32_t* lut256);
/* freq [fft_size/2+1] bin frequencies, targets [height] screen-row frequencies (Frequency_Resampler.xscaled).  A changed
 * height Fourier-resamples the carried column (Online_Linear_2D_resampler.set_height). */
int frt_specgram_set_screen(frt_specgram* h, const double* freq, const double* targets, int height);
/* Online_Linear_2D_resampler.set_ratio: STFT rate / pixel rate as the two numbers the widget passes */
int frt_specgram_set_ratio(frt_specgram* h, double interp_factor_L, double decim_factor_M);
/* chunk: n new samples (host, float64).  pixels_out: [height][max_cols] uint32 (host or device), row 0 = highest frequency;
 * *n_cols_out columns are written (often 0), *n_frames_out (nullable) spectra were computed. */
int frt_specgram_push(frt_specgram* h, const double* chunk, int n, uint32_t* pixels_out, int max_cols, int* n_cols_out,
                      int* n_frames_out);
int frt_specgram_reset(frt_specgram* h);

/* ---- K5: GCC-PHAT cross-correlation and the delay read-out --------------------------------------
 * frt_gcc_phat replaces generalized_cross_correlation (friture/signal/correlation.py:24-43): mean
 * removal, numpy.hanning window, two real FFTs, conj(D0) D1, PHAT weight 1/(1e-10 max|G| + |G|),
 * inverse real FFT.  length must be even with length/2 = R * M2, R in {1, 2, 4}, M2 <= 6144 and
 * 5-smooth (the default window of 24000 samples, friture/delay_estimator.py:114-115, qualifies).
 * The reference subtracts the means in place on its arguments; means_out lets a binding reproduce
 * that side effect. */
typedef struct frt_gcc frt_gcc;
int frt_gcc_create(frt_gcc** h, int length, int n_pairs);
void frt_gcc_destroy(frt_gcc* h);
int frt_gcc_set_stream(frt_gcc* h, void* hip_stream);
/* d0, d1, xcorr_out: [n_pairs][length] doubles; argmax_out: [n_pairs] index of max |xcorr| or NULL;
 * means_out: [n_pairs][2] or NULL. */
int frt_gcc_phat(frt_gcc* h, const double* d0, const double* d1, double* xcorr_out, int* argmax_out, double* means_out);

/* Peak read-out of Delay_Estimator_Widget.handle_new_data (friture/delay_estimator.py:134-176). */
typedef struct frt_delay_readout {
    int argmax;           /* index of max |smoothed|                                   :141-142 */
    int correlation_pct;  /* int(100 x/(1+x)), x = (0.12 max(0, peak/(3 std) - 1))^3    :171-176 */
    double delay_ms;      /* 1e3 argmax / rate, minus the window when past its half     :148-152 */
    double distance_m;    /* delay * 340 m/s                                            :167-168 */
    double extremum;      /* smoothed[argmax] (its sign is the polarity)                :146     */
} frt_delay_readout;
/* smoothed_out = alpha * xcorr + (1 - alpha) * old_smoothed (or xcorr when old_smoothed is NULL),
 * [n_pairs][length]; readout: [n_pairs] host structs. */
int frt_gcc_readout(frt_gcc* h, const double* xcorr, const double* old_smoothed, double alpha, double sample_rate,
                    double delayrange_s, double* smoothed_out, frt_delay_readout* readout);

/* ---- K6: screen-space stages of the spectrogram, and block-wise exponential smoothing -------------
 * Stateless float64 kernels, one per block of the reference's Transform_Pipeline
 * (friture/signal/transform_pipeline.py:23-34); the stateful bookkeeping of the online resampler
 * (which pixel columns a pushed column produces) is scalar arithmetic that stays with the caller.
 * Matrices are row-major; data/out buffers may be host or device memory. */
/* P5 Frequency_Resampler.push (friture/signal/frequency_resampler.py:67-83):
 * out[h][c] = numpy.interp(targets[h], freq, data[:, c]); data [n_bins][n_cols], out [height][n_cols]. */
int frt_freq_resample(const double* freq, int n_bins, const double* targets, int height, const double* data,
                      int n_cols, double* out);
/* P6 linear_interp_2D (friture/signal/linear_interp.py:57-60) for all pixel columns a push emits:
 * out[h][p] = data[h][src_col[p]] * (1 - a[p]) + prev * a[p], prev = data[h][src_col[p]-1] or old[h]
 * for column 0; data [height][n_cols], out [height][n_out]. */
int frt_time_resample(const double* data, const double* old, int height, int n_cols, const int* src_col,
                      const double* a, int n_out, double* out);
/* Fourier resampling of `count` real vectors x [count][n] -> y [count][m] (friture/signal/scipy_resample.py:51-141 with
 * window = None, axis = the vector): the N = min(n, m) lowest frequencies of fft(x) are kept, y = ifft(Y) * m / n.  Any
 * lengths (screen heights are arbitrary integers).  Used by Online_Linear_2D_resampler.set_height
 * (friture/signal/online_linear_2D_resampler.py:45-55) for the carried column when the plot is resized. */
int frt_fourier_resample(const double* x, int n, int count, double* y, int m);
/* P7 Color_Transform.push (friture/signal/color_tranform.py:48-51): out[i] = lut[int(clip(v[i],0,1)*255)] */
int frt_colour_map(const uint32_t* lut256, const double* values, int64_t count, uint32_t* out);
/* P5 + P6 + P7 in one call: Transform_Pipeline.push (friture/signal/transform_pipeline.py:29-34) over the spectrogram's three
 * blocks (friture/spectrogram.py:62-68) — np.interp onto the screen rows (frequency_resampler.py:67-83), the online time
 * resampler's lerp against its carried column (online_linear_2D_resampler.py:61-97, linear_interp.py:47-60), clip + LUT
 * (color_tranform.py:48-51) — the reference's operations in the reference's order, host arrays in and out.
 * norm [n_cols][nb] frame-major; freq [nb], targets [height]: Frequency_Resampler's freq and xscaled; old_in / old_out
 * [height]: the time resampler's carried column before / after; src, a [n_out]: its (source column, weight) pairs;
 * pixels_out [height][n_out] uint32, row 0 = lowest frequency. */
int frt_screen_columns(const double* norm, int nb, int n_cols, const double* freq, const double* targets, int height,
                       const double* old_in, const int* src, const double* a, int n_out, const uint32_t* lut256,
                       uint32_t* pixels_out, double* old_out);
/* P8 exp_smoothed_value_2d (friture/signal/exp_smoothing.py:91-107); nf = 1 gives exp_smoothed_value:
 * out[r] = alpha * dot(data[r][:n], kernel[nk-n:]) + previous[r] * (1-alpha)^n, n = min(nt, nk)
 * (previous is dropped when nt > nk). */
int frt_exp_smooth_2d(const double* kernel, int nk, double alpha, const double* data, int nf, int nt, int64_t row_stride,
                      const double* previous, double* out);
/* The octave-spectrum widget's smoothing of one chunk in ONE call (friture/octavespectrum.py:103-112 calls exp_smoothed_value
 * once per band, friture/signal/exp_smoothing.py:40-56): group g = the nf[g] bands of one octave, which share kernel
 * (kernels[g], nk[g] taps), alpha and length nt[g]; data[g] points at the group's rows (row_stride[g] doubles apart — the packed
 * band signals frt_octbank_filter returns are laid out like this).  square != 0: every datum is squared first (the widget smooths
 * y^2).  previous / out: one value per row, groups concatenated.  Every row equals its own frt_exp_smooth_2d call bit for bit.
 * Host arrays only (one chunk of the widget). */
int frt_exp_smooth_groups(int n_groups, const double* const* kernels, const int* nk, const double* alphas,
                          const double* const* data, const int* nf, const int* nt, const int64_t* row_stride, int square,
                          const double* previous, double* out);

/* ---- spectrum widget post-processing (Spectrum_Widget.handle_new_data, friture/spectrum.py:156-182) --
 * psd: the n_frames new PSD frames [n_frames][frame_stride] (float when psd_is_f32, else double; the
 * frame-major slab frt_stft_run writes).  smoothed = exp_smoothed_value_2d(kernel, alpha, psd^T, previous);
 * db = 10 log10(smoothed + 1e-30) + weight_db (or - 10 log10(ref_smoothed + 1e-30) in dual-channel mode);
 * *peak_index_out = argmax(db); *pitch_index_out = argmax of the harmonic product spectrum
 * s[:K] s[::2][:K] s[::3][:K], K = n_bins / 3, of smoothed (of ref_smoothed in dual-channel mode).
 * psd / previous / weight_db / ref_smoothed / smoothed_out: all host or all device; db_out may be a host array either way.
 * STREAM ORDER of device-resident arguments (this call and the other stateless entry points that accept device pointers:
 * frt_freq_resample, frt_time_resample, frt_colour_map, frt_exp_smooth_2d, frt_screen_columns): the kernel is launched on
 * the NULL stream, which is ordered behind every BLOCKING stream of the process and behind nothing else.  Whatever
 * produced `psd` must therefore have been enqueued on the null stream or a blocking stream (the default of
 * frt_stft_create; torch's current stream unless the caller made it a non-blocking side stream) — or be complete
 * (hipStreamSynchronize / an event wait) before the call.  A handle whose stream was set to a hipStreamNonBlocking one with
 * frt_stft_set_stream gives no such order. */
int frt_spectrum_post(const void* psd, int psd_is_f32, int n_frames, int n_bins, int64_t frame_stride,
                      const double* kernel, int nk, double alpha, const double* previous, const double* weight_db,
                      const double* ref_smoothed, double* smoothed_out, double* db_out, int* peak_index_out,
                      int* pitch_index_out);

/* ---- T1: pitch tracker (PitchTracker.estimate_pitch / update, friture/pitch_tracker.py:313-428) ---------
 * Per frame of fft_size samples (hop apart): |rfft(frame * hann)| -> np.interp onto the log-spaced grid
 * log_freqs[n_log] -> divide by its RMS -> strengths = kernels[n_candidates][n_log] @ grid spectrum ->
 * arg-max, parabolic vertex (fastParabolicInterp :160-193), index -> Hz, confidence = strength / 2.56,
 * frame level 20 log10(rms + eps); then the sequential gate: unvoiced (NaN) when level < min_db,
 * confidence < conf, or the estimate jumps more than p_delta semitones from the previous voiced one.
 * The grid and the kernel matrix are the caller's tables (_init_swipe :334-355, calcCosineKernel :195-264);
 * the gate's "previous estimate" is carried in the handle, per channel, across calls. */
typedef struct frt_pitch frt_pitch;
int frt_pitch_create(frt_pitch** h, int fft_size, int hop, int n_channels, double sample_rate, const double* log_freqs,
                     int n_log, const double* kernels, int n_candidates, double min_db, double conf, double p_delta);
void frt_pitch_destroy(frt_pitch* h);
int frt_pitch_set_stream(frt_pitch* h, void* hip_stream);
/* forget the previous estimates (prev_f0 = None, :292) */
int frt_pitch_reset(frt_pitch* h);
/* the gate's carried state: previous[n_channels] (host or device), NaN = no previous estimate */
int frt_pitch_set_previous(frt_pitch* h, const double* previous);
int frt_pitch_get_previous(frt_pitch* h, double* previous);
/* the widget changes the thresholds at run time (pitch_tracker.py:142-146) */
int frt_pitch_set_gate(frt_pitch* h, double min_db, double conf, double p_delta);
/* device scratch (spectra, grid spectra, strengths) a call may hold; longer runs go through it in chunks
 * of frames.  Default 1 GiB. */
int frt_pitch_set_scratch_limit(frt_pitch* h, int64_t bytes);
int64_t frt_pitch_frames_for(const frt_pitch* h, int64_t T);
/* x: [n_channels][x_stride] doubles, T valid samples per channel.  f0_out: [n_channels][n_frames], NaN =
 * unvoiced.  raw_out (or NULL): [3][n_channels][n_frames] = estimate before gating, confidence, dBFS. */
int frt_pitch_track(frt_pitch* h, const double* x, int64_t T, int64_t x_stride, double* f0_out, double* raw_out,
                    int64_t* n_frames_out);

#ifdef __cplusplus
}
#endif
#endif /* FRITURE_HIP_H */
