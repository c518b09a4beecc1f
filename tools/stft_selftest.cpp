// stft_selftest.cpp — standalone driver for the K1 entry points of libfriture_hip.so.
//
//   stft_selftest check            parity of every FFT size / hop class / output kind against a
//                                  double-precision host FFT (not the oracle: a quick sanity gate
//                                  that runs without Python)
//   stft_selftest bench [N hop C log2T kind run iters precision split]
//                                  HIP-event timing of frt_stft_run on device-resident input
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <time.h>

#include "friture_hip.h"

#define CK(x)                                                                   \
    do {                                                                        \
        int rc__ = (x);                                                         \
        if (rc__ != 0) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, frt_last_error()); \
            exit(2);                                                            \
        }                                                                       \
    } while (0)
#define HK(x)                                                                         \
    do {                                                                              \
        hipError_t e__ = (x);                                                         \
        if (e__ != hipSuccess) {                                                      \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e__));           \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

using cd = std::complex<double>;

static void host_fft(std::vector<cd>& a) {  // iterative radix-2, forward
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                cd w = std::polar(1.0, ang * (double)k);
                cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

static std::vector<double> host_psd(const double* x, int N) {
    std::vector<cd> a(N);
    for (int n = 0; n < N; ++n) a[n] = x[n] * 0.5 * (1.0 - std::cos(2.0 * M_PI * n / (N - 1)));
    host_fft(a);
    std::vector<double> p(N / 2 + 1);
    for (int k = 0; k <= N / 2; ++k) p[k] = std::norm(a[k]) / ((double)N * N);
    return p;
}

static int check_case(int N, int hop, int C, int frames, int precision, bool device_side, int run, bool split = false) {
    const int nb = N / 2 + 1;
    const int64_t T = (int64_t)N + (int64_t)hop * (frames - 1) + 3;  // a few unused tail samples
    const int64_t stride = T + (T & 1);
    std::mt19937 rng(1234 + N + hop);
    std::normal_distribution<double> nd(0.0, 0.25);
    std::vector<double> xd((size_t)C * stride);
    for (auto& v : xd) v = nd(rng);
    // a tone so that the spectrum has dynamic range
    for (int c = 0; c < C; ++c)
        for (int64_t t = 0; t < T; ++t) xd[c * stride + t] += 0.5 * std::sin(2 * M_PI * (1000.0 + 500 * c) * t / 48000.0);
    std::vector<float> xf(xd.size());
    for (size_t i = 0; i < xd.size(); ++i) {
        xf[i] = (float)xd[i];
        if (precision == 32) xd[i] = xf[i];
    }

    frt_stft* h = nullptr;
    CK(frt_stft_create(&h, N, hop, C, precision));
    CK(frt_stft_set_run_length(h, run));
    std::vector<double> wdb(nb);
    for (int k = 0; k < nb; ++k) wdb[k] = -3.0 + 0.001 * k;
    std::vector<uint32_t> lut(256);
    for (int i = 0; i < 256; ++i) lut[i] = 0xFF000000u | (i << 16) | ((255 - i) << 8) | (i ^ 0x55);
    CK(frt_stft_set_epilogue(h, wdb.data(), -140.0, 0.0, lut.data()));

    const int64_t F = frt_stft_frames_for(h, T);
    if (F != frames) {
        fprintf(stderr, "frame count %lld != %d\n", (long long)F, frames);
        return 1;
    }
    const size_t esz = precision == 32 ? 4 : 8;
    const void* xin = precision == 32 ? (const void*)xf.data() : (const void*)xd.data();
    std::vector<char> out[4];
    for (int kind = 0; kind < 4; ++kind) {
        const size_t oesz = kind == FRT_STFT_IMAGE ? 4 : esz;
        out[kind].assign((size_t)C * F * nb * oesz, 0x7f);
        int64_t nf = 0;
        if (split) {
            // split rows (frt_stft_run_split): rows [C][F][N/2] + Nyquist plane [C][F], reassembled into the packed shape checked below
            const size_t row_bytes = (size_t)C * F * (nb - 1) * oesz, nyq_bytes = (size_t)C * F * oesz;
            std::vector<char> rows(row_bytes, 0x7f), nyq(nyq_bytes, 0x7f);
            if (device_side) {
                void *dx, *drows, *dnyq;
                HK(hipMalloc(&dx, xd.size() * esz));
                HK(hipMalloc(&drows, row_bytes));
                HK(hipMalloc(&dnyq, nyq_bytes));
                HK(hipMemcpy(dx, xin, xd.size() * esz, hipMemcpyHostToDevice));
                CK(frt_stft_run_split(h, kind, dx, T, stride, drows, dnyq, &nf));
                HK(hipDeviceSynchronize());
                HK(hipMemcpy(rows.data(), drows, row_bytes, hipMemcpyDeviceToHost));
                HK(hipMemcpy(nyq.data(), dnyq, nyq_bytes, hipMemcpyDeviceToHost));
                HK(hipFree(dx));
                HK(hipFree(drows));
                HK(hipFree(dnyq));
            } else {
                CK(frt_stft_run_split(h, kind, xin, T, stride, rows.data(), nyq.data(), &nf));
            }
            for (size_t r = 0; r < (size_t)C * F; ++r) {
                memcpy(&out[kind][r * nb * oesz], &rows[r * (nb - 1) * oesz], (nb - 1) * oesz);
                memcpy(&out[kind][(r * nb + nb - 1) * oesz], &nyq[r * oesz], oesz);
            }
        } else if (device_side) {
            void *dx, *dout;
            HK(hipMalloc(&dx, xd.size() * esz));
            HK(hipMalloc(&dout, out[kind].size()));
            HK(hipMemcpy(dx, xin, xd.size() * esz, hipMemcpyHostToDevice));
            HK(hipMemset(dout, 0x7f, out[kind].size()));
            CK(frt_stft_run(h, kind, dx, T, stride, dout, &nf));
            HK(hipDeviceSynchronize());
            HK(hipMemcpy(out[kind].data(), dout, out[kind].size(), hipMemcpyDeviceToHost));
            HK(hipFree(dx));
            HK(hipFree(dout));
        } else {
            CK(frt_stft_run(h, kind, xin, T, stride, out[kind].data(), &nf));
        }
        if (nf != F) return 1;
    }
    auto get = [&](int kind, size_t idx) -> double {
        return precision == 32 ? (double)((float*)out[kind].data())[idx] : ((double*)out[kind].data())[idx];
    };
    double worst = 0, worst_db = 0, worst_norm = 0;
    long pix_bad = 0, pix_edge = 0;
    for (int c = 0; c < C; ++c)
        for (int64_t f = 0; f < F; ++f) {
            std::vector<double> ref = host_psd(&xd[c * stride + f * hop], N);
            double mx = 0, err = 0;
            for (int k = 0; k < nb; ++k) {
                const size_t idx = ((size_t)c * F + f) * nb + k;
                mx = std::max(mx, ref[k]);
                err = std::max(err, std::fabs(get(0, idx) - ref[k]));
                const double db = 10.0 * std::log10(ref[k] + 1e-30) + wdb[k];
                // dB of a bin far below the frame maximum is ill-conditioned in fp32; gate bins
                // within 60 dB of the maximum only
                const double nrm = (db + 140.0) / 140.0;
                if (ref[k] > 1e-6 * 1.0 * mx || precision == 64) {
                    if (ref[k] > 1e-6 * mx) {
                        worst_db = std::max(worst_db, std::fabs(get(1, idx) - db));
                        worst_norm = std::max(worst_norm, std::fabs(get(2, idx) - nrm));
                    }
                }
                const double cl = std::min(std::max(nrm, 0.0), 1.0);
                const uint32_t want = lut[(int)(cl * 255.0)];
                const uint32_t got = ((uint32_t*)out[3].data())[idx];
                if (want != got) {
                    const double fr = cl * 255.0 - std::floor(cl * 255.0);
                    if (fr < 1e-2 || fr > 1 - 1e-2 || ref[k] < 1e-6 * mx) ++pix_edge; else ++pix_bad;
                }
            }
            worst = std::max(worst, err / mx);
        }
    frt_stft_destroy(h);
    const double tol = precision == 32 ? 1e-5 : 1e-12;
    const double tol_db = precision == 32 ? 2e-2 : 1e-9;
    const bool ok = worst <= tol && worst_db <= tol_db && pix_bad == 0;
    printf("%s N=%5d hop=%5d C=%d F=%3d p%d %s%s run=%2d  psd_relmax=%.3e dB_abs=%.3e norm_abs=%.3e pix_bad=%ld pix_edge=%ld\n",
           ok ? "ok  " : "FAIL", N, hop, C, frames, precision, device_side ? "dev " : "host", split ? " split" : "", run, worst, worst_db,
           worst_norm, pix_bad, pix_edge);
    return ok ? 0 : 1;
}

static int do_check() {
    int fails = 0;
    for (int N = 32; N <= 16384; N *= 2) {
        fails += check_case(N, N / 2, 2, 21, 32, true, 0);
        fails += check_case(N, N / 4, 1, 19, 32, true, 5);
        fails += check_case(N, 3 * N / 8, 2, 9, 32, false, 4);   // generic even hop
        fails += check_case(N, N / 2 + 1, 1, 7, 32, true, 3);      // odd hop: scalar loads
        fails += check_case(N, N / 4, 1, 5, 64, false, 0);
    }
    for (int N = 32; N <= 1024; N *= 2) {                       // split rows: every N <= 1024 instance class, both sides, both precisions
        fails += check_case(N, N / 2, 2, 70, 32, true, 0, true);
        fails += check_case(N, N / 4, 1, 19, 32, false, 5, true);
        fails += check_case(N, N / 2 + 1, 1, 7, 32, true, 3, true);
        fails += check_case(N, N / 2, 1, 9, 64, true, 0, true);
    }
    fails += check_case(1024, 512, 3, 300, 32, true, 0);
    fails += check_case(1024, 512, 1, 1, 32, true, 0);
    printf("%s (%d failing cases)\n", fails ? "SELFTEST FAILED" : "SELFTEST OK", fails);
    return fails ? 1 : 0;
}

static int64_t nf_g;
static int do_bench(int N, int hop, int C, int log2T, int kind, int run, int iters, int precision, int split) {
    const int64_t T = 1ll << log2T;
    frt_stft* h = nullptr;
    CK(frt_stft_create(&h, N, hop, C, precision));
    CK(frt_stft_set_run_length(h, run));
    std::vector<uint32_t> lut(256);
    for (int i = 0; i < 256; ++i) lut[i] = 0xFF000000u | (i * 0x010101);
    CK(frt_stft_set_epilogue(h, nullptr, -140.0, 0.0, lut.data()));
    const int64_t F = frt_stft_frames_for(h, T);
    const int nb = N / 2 + 1;
    std::vector<float> x((size_t)C * T);
    std::mt19937 rng(42);
    std::normal_distribution<float> nd(0.f, 0.25f);
    for (auto& v : x) v = nd(rng);
    // FRT_BENCH_SETS=k: the timed launches rotate over k distinct input/output buffer pairs, so that no launch
    // finds its input in the 256 MB Infinity Cache (k = 1: the same batch every launch)
    const int sets = getenv("FRT_BENCH_SETS") ? atoi(getenv("FRT_BENCH_SETS")) : 1;
    const size_t esz = precision == 32 ? 4 : 8, oesz = kind == FRT_STFT_IMAGE ? 4 : esz;
    std::vector<double> xd;
    if (precision == 64) xd.assign(x.begin(), x.end());
    const void* xhost = precision == 32 ? (const void*)x.data() : (const void*)xd.data();
    char* dx0;
    char* dout0;
    // (each set's buffers start on 4 KB boundaries: the split rows are 64-byte lines only if their base is)
    const size_t out_bytes = ((size_t)C * F * nb * oesz + 4095) / 4096 * 4096, in_bytes = (x.size() * esz + 4095) / 4096 * 4096;
    const size_t row_bytes = (size_t)C * F * (nb - 1) * oesz;        // split layout: the rows, then the Nyquist plane, in the same allocation
    auto run_once = [&](const void* xin, char* o) {
        if (split) CK(frt_stft_run_split(h, kind, xin, T, T, o, o + row_bytes, &nf_g));
        else CK(frt_stft_run(h, kind, xin, T, T, o, &nf_g));
    };
    // FRT_BENCH_OUT_SHIFT / FRT_BENCH_IN_SHIFT (bytes, multiples of 4096): the buffers start that far into their allocations — where the
    // rows lie relative to the samples in the memory's channel interleave (tools/exp/session_r5t.sh)
    const size_t out_shift = getenv("FRT_BENCH_OUT_SHIFT") ? (size_t)atoll(getenv("FRT_BENCH_OUT_SHIFT")) : 0;
    const size_t in_shift = getenv("FRT_BENCH_IN_SHIFT") ? (size_t)atoll(getenv("FRT_BENCH_IN_SHIFT")) : 0;
    char *dx_alloc, *dout_alloc;
    HK(hipMalloc(&dx_alloc, in_bytes * sets + in_shift));
    HK(hipMalloc(&dout_alloc, out_bytes * sets + out_shift));
    dx0 = dx_alloc + in_shift;
    dout0 = dout_alloc + out_shift;
    if (getenv("FRT_BENCH_SHOW_PTRS")) printf("x %p out %p in_bytes %zu out_bytes %zu\n", (void*)dx0, (void*)dout0, in_bytes, out_bytes);
    for (int k = 0; k < sets; ++k) HK(hipMemcpy(dx0 + in_bytes * k, xhost, x.size() * esz, hipMemcpyHostToDevice));
    char* dx = dx0;
    void* dout = dout0;
    hipStream_t s;
    HK(hipStreamCreate(&s));
    CK(frt_stft_set_stream(h, s));
    int64_t nf;
    // Clock ramp: after idle the GPU needs tens of milliseconds of continuous work to reach its sustained
    // clocks (a 50-launch measurement straight after start-up reads ~17 % slow).  Pre-warm for FRT_BENCH_PREWARM_MS
    // (default 250 ms) of back-to-back launches before anything is timed.
    {
        const double prewarm_ms = getenv("FRT_BENCH_PREWARM_MS") ? atof(getenv("FRT_BENCH_PREWARM_MS")) : 250.0;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        int k = 0;
        for (;;) {
            for (int i = 0; i < 64; ++i, ++k) run_once(dx0 + in_bytes * (k % sets), dout0 + out_bytes * (k % sets));
            HK(hipStreamSynchronize(s));
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 >= prewarm_ms) break;
        }
    }
    for (int i = 0; i < 3; ++i) run_once(dx, (char*)dout);
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    HK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) run_once(dx0 + in_bytes * (i % sets), dout0 + out_bytes * (i % sets));
    HK(hipEventRecord(e1, s));
    HK(hipEventSynchronize(e1));
    float ms = 0;
    HK(hipEventElapsedTime(&ms, e0, e1));
    // second measurement: one event pair per launch with the stream drained and a host pause between
    // launches (what a serialising profiler sees) to separate kernel time from sustained-load effects
    double iso_ms = 0;
    for (int i = 0; i < iters; ++i) {
        HK(hipStreamSynchronize(s));
        struct timespec ts = {0, 300000};
        nanosleep(&ts, nullptr);
        HK(hipEventRecord(e0, s));
        run_once(dx, (char*)dout);
        HK(hipEventRecord(e1, s));
        HK(hipEventSynchronize(e1));
        float m1 = 0;
        HK(hipEventElapsedTime(&m1, e0, e1));
        iso_ms += m1;
    }
    iso_ms /= iters;
    const double per = ms / iters * 1e-3;
    const double spectra = (double)C * F / per;
    const double bytes = (double)C * F * ((double)esz * hop + (double)oesz * nb);
    printf("bench p%d N=%d hop=%d C=%d T=2^%d F=%lld kind=%d run=%d sets=%d %s: %.3f ms/launch  %.4e spectra/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)  [isolated launch: %.3f ms]\n",
           precision, N, hop, C, log2T, (long long)F, kind, run, sets, split ? "split" : "packed", per * 1e3, spectra, bytes / per * 1e-9, bytes / per / 8e12 * 100, iso_ms);
    frt_stft_destroy(h);
    HK(hipFree(dx_alloc));
    HK(hipFree(dout_alloc));
    return 0;
}

int main(int argc, char** argv) {
    int ncu = 0;
    int64_t hbm = 0;
    CK(frt_init(0, &ncu, &hbm));
    printf("%s: %d CUs, %.1f GB HBM\n", frt_version(), ncu, hbm / 1e9);
    if (argc < 2 || !strcmp(argv[1], "check")) return do_check();
    if (!strcmp(argv[1], "bench")) {
        int N = argc > 2 ? atoi(argv[2]) : 1024;
        int hop = argc > 3 ? atoi(argv[3]) : N / 2;
        int C = argc > 4 ? atoi(argv[4]) : 1;
        int log2T = argc > 5 ? atoi(argv[5]) : 26;
        int kind = argc > 6 ? atoi(argv[6]) : 0;
        int run = argc > 7 ? atoi(argv[7]) : 0;
        int iters = argc > 8 ? atoi(argv[8]) : 20;
        int precision = argc > 9 ? atoi(argv[9]) : 32;
        int split = argc > 10 ? atoi(argv[10]) : 0;
        return do_bench(N, hop, C, log2T, kind, run, iters, precision, split);
    }
    fprintf(stderr, "usage: stft_selftest check | bench [N hop C log2T kind run iters precision split]\n");
    return 2;
}
