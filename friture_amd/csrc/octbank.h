// octbank.h — the octave-bank handle shared by the exact IIR path (iir.hip) and the FFT
// overlap-add path (ola.hip).
#pragma once
#include "common.h"
#include "fft_mixed.h"

namespace frt {
constexpr int kStates = 16;        // one DPP row per filter
constexpr int kMaxOrder = 15;
constexpr int kCoefStride = 2 * (kMaxOrder + 1);   // b[0..15], a[0..15]
constexpr int kMaxFilters = 25;    // 24 band-passes + decimator
constexpr int kNOctave = 9;        // friture/filter.py:7
constexpr int kFirLength = 512;    // friture/octavefilters.py:35
constexpr int kTail = kFirLength - 1;

inline void stage_lengths(int n, int* len) {
    len[0] = n;
    for (int j = 1; j < kNOctave; ++j) len[j] = (len[j - 1] + 1) / 2;     // x[::2]
}
}  // namespace frt

// overlap-add state of mode 1 (ola.hip)
struct frt_ola_state {
    int fft_size[frt::kNOctave];
    frt::MixedPlan plan[frt::kNOctave];
    frt::DeviceBuffer tw[frt::kNOctave], twl[frt::kNOctave], H[frt::kNOctave];
    frt::DeviceBuffer pending;          // [9][C][nfilt][511]
    // batched path (ola.hip, ola_batch_kernel): one transform size for every stage, tables built at the first batched call;
    // its launches read the tails of `pending` and write the new ones to `pending_next`, then the two swap
    frt::DeviceBuffer pending_next;
    frt::DeviceBuffer btw, btwl, bH, bHw, ewt;
    frt::DeviceBuffer multi_tab[2];     // argument tables of ola_pair_multi_kernel, one per parity of the tails' swap
    std::vector<char> multi_host[2];    // what each holds
    void* multi_pin[2] = {nullptr, nullptr};   // page-locked staging of the tables' uploads (an async copy keeps its source pointer)
    int multi_parity = 0;
    std::vector<long long> ewt_off;     // per band: offset of its smoothing weights in ewt
    int ewt_block = 0;
    std::vector<double> ewt_alpha;
    std::vector<double> h_taps;         // [nfilt][512] kept for the lazily built tables
    // chunk path (ola.hip, ola_chunk_*_kernel): the taps themselves and the per-band weight offsets on the device
    frt::DeviceBuffer taps, ewt_off_dev, xs;
    int ewt_n = -1;                     // chunk length the `whole` weights were built for
};

struct frt_octbank {
    int bpo = 0, n_channels = 0, mode = 0, nbands = 0, nfilt = 0;
    int chunk0 = 0;                         // 0 = sequential (bit exact); else samples per chunk at octave 0
    hipStream_t stream = nullptr;
    std::vector<double> h_coef;             // [nfilt][kCoefStride]
    std::vector<int> h_order;
    frt::DeviceBuffer coef, order, state;        // state: [9][C][nfilt][16]
    frt::DeviceBuffer xin, ypacked, xbuf[frt::kNOctave], chunk_end, chunk_init, power;
    frt::DeviceBuffer state_snap;                         // [9][C][nfilt][16] carried states as a stage found them (look-back output pass)
    std::vector<int> slook;                               // per stage: chunks the filters' decay spans
    std::vector<int> sgroup, shalo;                       // per stage: chunks per scan row, rows the filters' decay spans (iir_scan_kernel)
    frt::DeviceBuffer eseg;                               // per-split carries of the block-energy recurrence (long batches)
    frt::DeviceBuffer zs_table, zs_table_m, zs_rowmap;   // zero-state response tables of the time-parallel mode (iir.hip)
    std::vector<size_t> zs_offset;           // per stage, in doubles
    int zs_rows = 0, zs_rows_padded = 0;
    bool zero_state_by_recurrence = false;   // A/B and tests: pass 1 as a second run of the recurrence
    frt::DeviceBuffer eblock, alpha, decay_n, smooth, weight, eout;
    std::vector<double> alpha_host, decay_host, weight_host;     // what the three tables hold (upload_if_changed)
    int power_chunk0 = -1;
    int power_n = -1;
    int zs_dec_tiles = 0, zs_band_tiles = 0;   // row tiles (of 16) of the zero-state table: the decimator's first, then the band filters'
    // the time-parallel mode's side streams: the band filters of a stage run beside the decimator chain (iir.hip, run_stages)
    static constexpr int kSideStreams = 3;
    hipStream_t side[kSideStreams] = {};
    hipEvent_t ev_start = nullptr, ev_x[frt::kNOctave + 1] = {}, ev_side[kSideStreams] = {};
    frt_ola_state* ola = nullptr;
    // interactive host-buffer path: the per-block launch sequence (H2D, nine stage kernels, D2H) is
    // launch bound, so it is captured once per block length into a hipGraph and replayed
    struct StreamGraph {
        int n = 0;
        hipGraphExec_t exec = nullptr;
        const void* ptrs[2 + frt::kNOctave] = {};     // device buffers baked into the graph
    };
    std::vector<StreamGraph> graphs;
    hipStream_t gstream = nullptr;
    void* pin_in = nullptr;
    void* pin_out = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0;
    int warmed_n = -1;
    bool use_graph = true;
    size_t stage_state_elems() const { return (size_t)n_channels * nfilt * frt::kStates; }
};


// implemented in ola.hip
int frt_ola_create(frt_octbank* h, const double* boct_fir, const double* bdec_fir);
void frt_ola_destroy(frt_octbank* h);
int frt_ola_reset(frt_octbank* h);
int frt_ola_filter(frt_octbank* h, const double* d_x, int n, double* d_y, int64_t y_cstride);
// batched: x [C][n] (float when x_f32) as if fed in blocks of <= 1024 samples; band signals to d_y (nullable) and / or
// zero-state block energies for blocks of eblock0 input samples to d_eblock [C][nblocks][nbands] (nullable)
int frt_ola_filter_batch(frt_octbank* h, const void* d_x, int x_f32, int64_t n, double* d_y, int64_t y_cstride,
                         double* d_eblock, int eblock0, int nblocks, const double* alphas);
// one chunk of 1..1024 samples = one energy block (the octave-spectrum widget's handler): the smoothed band energies of
// x [C][n] (device-accessible memory: HBM or pinned host) straight to `out` [C][nbands] (float when out_f32; device-accessible),
// two launches; decay_n / smooth / weight_db: the handle's device tables (weight_db nullable)
// ... and the band signals of such a block: x [C][n] float64, y packed [C][y_cstride], both device accessible
int frt_ola_chunk_filter(frt_octbank* h, const double* x, int n, double* y, int64_t y_cstride);
int frt_ola_chunk_energies(frt_octbank* h, const void* x, int x_f32, int n, const double* alphas, const double* d_decay_n,
                           double* d_smooth, const double* d_weight_db, int as_db, void* out, int out_f32);
