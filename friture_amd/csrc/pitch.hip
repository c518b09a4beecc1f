// pitch.hip — T1: SWIPE-style pitch tracker (friture/pitch_tracker.py:334-428), float64.
//
// Per frame the reference computes |rfft(frame * hann)|, resamples it onto a log-spaced frequency
// grid (np.interp), normalises by its RMS, multiplies by a [candidates x grid] kernel matrix, takes
// the arg-max candidate, refines it with a parabola and gates the result (level, confidence, jump).
// Here a batch of frames goes through five launches:
//   1. K1 (stft.hip, float64 instance) -> power spectra P[frame][N/2+1]
//   2. pitch_loggrid_kernel   P -> S: magnitude, interpolation, RMS normalisation;  pitch_level_kernel: frame level (dBFS)
//   3. pitch_strength_kernel  strengths = kernels x S   — the one dense contraction of the path:
//      [481 x 1023] x [1023 x frames] in float64.  MI355X has no float64 matrix rate above its vector
//      rate, so this is a register-tiled FMA kernel: a wavefront owns 128 candidates x 16 frames
//      (2 x 16 accumulators per lane); the kernel matrix is stored transposed (grid-major) so that a
//      wavefront's loads are contiguous, and S is stored [frame/8][grid][frame%8] so that eight frame
//      values of a grid point are ONE wave-uniform 64-byte scalar load feeding the FMAs from SGPRs.
//   4. pitch_pick_kernel      arg-max, parabolic vertex, index -> Hz, confidence
//   5. pitch_gate_kernel      the sequential voiced/unvoiced decision, one thread per channel
// Compiled with -ffp-contract=off: everything outside the contraction keeps the reference's
// operation order; the contraction itself uses explicit fma().
#include <cmath>
#include <cstdlib>
#include <limits>

#include "common.h"

namespace frt {

constexpr int kFramesPerGroup = 8;      // frames sharing one scalar load of S
constexpr int kFramesPerWave = 8;       // frames a wavefront accumulates (one group)
constexpr int kFramesPerBlock = 32;     // 4 wavefronts x 8 frames
constexpr int kCandPerLane = 4;
constexpr int kCandPerWave = 256;       // 64 lanes x 4 candidates

struct PitchArgs {
    const double* x;          // [C][x_stride]
    long long x_stride;
    const double* psd;        // [C*Fc][nb]
    double* s;                // [(C*Fc)/8][Lp][8]
    double* strength;         // [C*Fc][Kp]
    double* raw;              // [3][C][F]: f0 before gating, confidence, dBFS
    const double* freqs;      // [L]
    const int* jidx;          // [L] left bin of each grid frequency
    const double* kt;         // [L][Kp] transposed kernel matrix, zero padded
    int N, nb, hop, L, Lp, K, Kp, C;
    long long F, f_start, Fc;  // frames per channel in total / first frame of this chunk / frames in this chunk
    double binw;              // sample_rate / N
};

__device__ __forceinline__ double wave_sum(double v) {
    // fixed-order butterfly: every lane ends with the same, scheduling-independent sum
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// |rfft| on the log grid: np.interp(freqs, k * binw, |X|)  (pitch_tracker.py:370-378)
__device__ __forceinline__ double grid_value(const PitchArgs& a, const double* P, int l) {
    const int j = a.jidx[l];
    const double xq = a.freqs[l];
    const double m0 = sqrt(P[j]) * (double)a.N;
    const double x0 = (double)j * a.binw;
    if (j >= a.nb - 1 || x0 == xq) return m0;
    const double m1 = sqrt(P[j + 1]) * (double)a.N;
    const double slope = (m1 - m0) / ((double)(j + 1) * a.binw - x0);
    return slope * (xq - x0) + m0;
}

// One workgroup per group of 8 frames; lane = 8 * (grid point mod 8) + frame, so that a store instruction
// covers 8 grid points x 8 frames = 512 contiguous bytes of the [group][grid][8] layout; the four wavefronts
// split the grid.  Pass 1 writes the unnormalised grid spectrum and sums its squares, pass 2 divides the lane's own
// values by the frame's RMS — 256 bytes of LDS, one barrier, occupancy bounded by registers only.
__global__ void __launch_bounds__(256) pitch_loggrid_kernel(const PitchArgs a) {
    __shared__ double part[4][kFramesPerGroup];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long group = blockIdx.x;
    const long long total = (long long)a.C * a.Fc;
    const int q = lane & 7, li = lane >> 3;
    const long long gf = group * kFramesPerGroup + q;
    double* out = a.s + group * (long long)a.Lp * kFramesPerGroup;
    const bool live = gf < total;
    const double* P = a.psd + (live ? gf : 0) * a.nb;
    const int lq = a.Lp / 4, l0 = wave * lq, l1 = l0 + lq;      // Lp is a multiple of 32
    // pass 1: the unnormalised grid spectrum goes to its final place, the squares are summed
    double ss = 0.0;
    for (int l = l0 + li; l < l1; l += 8) {
        double v = 0.0;
        if (live && l < a.L) {
            v = grid_value(a, P, l);
            ss += v * v;
        }
        out[l * kFramesPerGroup + q] = v;
    }
    // the eight lanes of a frame hold interleaved partial sums; fixed-order butterfly over lane bits 3..5
    for (int o = 8; o < 64; o <<= 1) ss += __shfl_xor(ss, o, 64);
    if (li == 0) part[wave][q] = ss;
    __syncthreads();
    ss = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
    const double rms = sqrt(ss / (double)a.L);                      // :379
    // pass 2: every lane divides the values it wrote itself (:380; 0/0 = nan for silence, as upstream)
    if (live)
        for (int l = l0 + li; l < l1 && l < a.L; l += 8) out[l * kFramesPerGroup + q] = out[l * kFramesPerGroup + q] / rms;
}

// The same kernel for grids of 32*VPL points (the widget's grid has 1023 -> Lp = 1024, VPL = 32): a lane's VPL values
// stay in registers between the two passes, so S is written once and never read back, and the fully unrolled
// pass 1 lets the compiler batch the index/bin loads of many grid points.  Operation for operation the same
// arithmetic as above (bit-identical results).
template <int VPL>
__global__ void __launch_bounds__(256) pitch_loggrid_reg_kernel(const PitchArgs a) {
    __shared__ double part[4][kFramesPerGroup];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long group = blockIdx.x;
    const long long total = (long long)a.C * a.Fc;
    const int q = lane & 7, li = lane >> 3;
    const long long gf = group * kFramesPerGroup + q;
    double* out = a.s + group * (long long)a.Lp * kFramesPerGroup;
    const bool live = gf < total;
    const double* P = a.psd + (live ? gf : 0) * a.nb;
    const int l0 = wave * (8 * VPL) + li;
    double v[VPL];
    double ss = 0.0;
    // branch-free in batches of 8 grid points: all index loads, then all bin loads, then the arithmetic of
    // grid_value() with its early-outs turned into selects (clamped indices keep every load in bounds)
    constexpr int B = 8;
    static_assert(VPL % B == 0, "batching");
#pragma unroll
    for (int i0 = 0; i0 < VPL; i0 += B) {
        int j[B];
        double xq[B], p0[B], p1[B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int lc = min(l0 + 8 * (i0 + i), a.L - 1);
            j[i] = a.jidx[lc];
            xq[i] = a.freqs[lc];
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            p0[i] = P[j[i]];
            p1[i] = P[min(j[i] + 1, a.nb - 1)];
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int l = l0 + 8 * (i0 + i);
            const double m0 = sqrt(p0[i]) * (double)a.N;
            const double m1 = sqrt(p1[i]) * (double)a.N;
            const double x0 = (double)j[i] * a.binw;
            const double slope = (m1 - m0) / ((double)(j[i] + 1) * a.binw - x0);
            const double r = slope * (xq[i] - x0) + m0;
            const double g = (j[i] >= a.nb - 1 || x0 == xq[i]) ? m0 : r;
            const bool on = live && l < a.L;
            v[i0 + i] = on ? g : 0.0;
            ss += on ? g * g : 0.0;
        }
    }
    for (int o = 8; o < 64; o <<= 1) ss += __shfl_xor(ss, o, 64);
    if (li == 0) part[wave][q] = ss;
    __syncthreads();
    ss = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
    const double rms = sqrt(ss / (double)a.L);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int l = l0 + 8 * i;
        out[l * kFramesPerGroup + q] = (live && l < a.L) ? v[i] / rms : 0.0;
    }
}

// frame level: 20 log10(sqrt(mean(frame^2)) + eps)   (:399-400).  One wavefront per frame.
__global__ void __launch_bounds__(256) pitch_level_kernel(const PitchArgs a) {
    const int lane = threadIdx.x & 63;
    const long long gf = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gf >= (long long)a.C * a.Fc) return;
    const int chan = (int)(gf / a.Fc);
    const long long f = a.f_start + (gf - (long long)chan * a.Fc);
    const double* xf = a.x + chan * a.x_stride + f * a.hop;
    double e = 0.0;
    for (int n = lane; n < a.N; n += 64) e += xf[n] * xf[n];
    e = wave_sum(e);
    if (lane == 0) {
        const double r = sqrt(e / (double)a.N);
        a.raw[(2ll * a.C + chan) * a.F + f] = 20.0 * log10(r + std::numeric_limits<double>::epsilon());
    }
}

// The same level when the hop divides the frame (every overlap the widgets offer): overlapping frames share
// hop-sized blocks, so the sum of squares of every block is formed once (one wavefront per block) ...
__global__ void __launch_bounds__(256) pitch_block_energy_kernel(const double* __restrict__ x, long long x_stride, int hop,
                                                                 long long first_block, long long n_blocks, int C,
                                                                 double* __restrict__ eb) {
    const int lane = threadIdx.x & 63;
    const long long gb = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gb >= n_blocks * C) return;
    const int chan = (int)(gb / n_blocks);
    const long long b = gb - (long long)chan * n_blocks;
    const double* xb = x + chan * x_stride + (first_block + b) * hop;
    double e = 0.0;
    for (int n = lane; n < hop; n += 64) e += xb[n] * xb[n];
    e = wave_sum(e);
    if (lane == 0) eb[gb] = e;
}

// ... and a frame adds up its N / hop blocks.
__global__ void __launch_bounds__(256) pitch_level_from_blocks_kernel(const PitchArgs a, const double* __restrict__ eb,
                                                                      long long n_blocks, int per_frame) {
    const long long gf = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gf >= (long long)a.C * a.Fc) return;
    const int chan = (int)(gf / a.Fc);
    const long long fl = gf - (long long)chan * a.Fc;                  // frame inside the chunk = its first block
    double e = 0.0;
    for (int j = 0; j < per_frame; ++j) e += eb[(long long)chan * n_blocks + fl + j];
    const double r = sqrt(e / (double)a.N);
    a.raw[(2ll * a.C + chan) * a.F + a.f_start + fl] = 20.0 * log10(r + std::numeric_limits<double>::epsilon());
}

// strengths[frame][cand] = sum_l kt[l][cand] * S[frame][l]    (pitch_tracker.py:383)
// A lane owns kCandPerLane candidates x kFramesPerWave frames.  Software pipeline over pairs of grid points: scalar
// loads return out of order, so ANY use of one needs lgkmcnt(0) — placed by hand right BEFORE the next pair's
// loads are issued (where only the current pair is outstanding), not where the compiler would put it (before the
// first use, i.e. after the next pair was requested, draining it as well).  The next pair's 64-byte S rows and
// matrix elements are then in flight during the 64 FMAs of the current pair.
__global__ void __launch_bounds__(256) pitch_strength_kernel(const double* __restrict__ kt, const double* __restrict__ s,
                                                             double* __restrict__ strength, const int* __restrict__ lrange,
                                                             int L, int Lp, int Kp, long long n_frames) {
    static_assert(kFramesPerWave == kFramesPerGroup && kCandPerLane == 4, "tile: 4 candidates x 8 frames per lane");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long long group = (long long)blockIdx.x * 4 + wave;                 // the 8-frame group of this wave
    const int c0 = blockIdx.y * kCandPerWave + kCandPerLane * lane;
    if (group * kFramesPerGroup >= n_frames) return;
    const double* __restrict__ sg = s + group * (long long)Lp * kFramesPerGroup;
    const double* __restrict__ kc = kt + c0;
    double acc[kCandPerLane][kFramesPerGroup];
#pragma unroll
    for (int i = 0; i < kCandPerLane; ++i)
#pragma unroll
        for (int q = 0; q < kFramesPerGroup; ++q) acc[i][q] = 0.0;

    struct Pair {                      // two consecutive grid points
        double2 k[2][2];               // [point][half]: the lane's four candidates
        double sv[2][kFramesPerGroup];
    };
    // No bounds in the loop: the matrix and every group's S rows are zero-padded to Lp (a multiple of 4) rows and
    // both buffers carry four spare rows, so the last iteration's look-ahead reads valid memory it never uses.
    auto fetch = [&](int l, Pair& p) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            p.k[u][0] = *(const double2*)(kc + (long long)(l + u) * Kp);
            p.k[u][1] = *(const double2*)(kc + (long long)(l + u) * Kp + 2);
#pragma unroll
            for (int q = 0; q < kFramesPerGroup; ++q) p.sv[u][q] = sg[(long long)(l + u) * kFramesPerGroup + q];
        }
    };
    auto use = [&](const Pair& p) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const double k[4] = {p.k[u][0].x, p.k[u][0].y, p.k[u][1].x, p.k[u][1].y};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < kFramesPerGroup; ++q) acc[i][q] = __builtin_fma(k[i], p.sv[u][q], acc[i][q]);
        }
    };
    constexpr int kLgkm0 = 0xC07F;     // s_waitcnt lgkmcnt(0), vmcnt / expcnt untouched
    // The kernel matrix is banded (a candidate's kernel is zero below a quarter of its frequency and above its
    // last harmonic: 26 % exact zeros on the widget's grid): the rows in which ALL 256 candidates of this block
    // are zero are skipped — exact, since S is finite (or NaN throughout, which any live row propagates).
    const int l_begin = lrange[2 * blockIdx.y], l_end = lrange[2 * blockIdx.y + 1];      // multiples of 4
    Pair a, b;
    fetch(l_begin, a);
    for (int l = l_begin; l < l_end; l += 4) {
        // sched_barrier: keep the machine scheduler from sinking the look-ahead loads down to their first use
        __builtin_amdgcn_s_waitcnt(kLgkm0);
        fetch(l + 2, b);
        __builtin_amdgcn_sched_barrier(0);
        use(a);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(kLgkm0);
        fetch(l + 4, a);
        __builtin_amdgcn_sched_barrier(0);
        use(b);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < kFramesPerGroup; ++q) {
        const long long gf = group * kFramesPerGroup + q;
        if (gf < n_frames) {
            double* o = strength + gf * Kp + c0;
            *(double2*)o = double2{acc[0][q], acc[1][q]};
            *(double2*)(o + 2) = double2{acc[2][q], acc[3][q]};
        }
    }
}

// np.argmax ordering: the first NaN wins, otherwise the first maximum
__device__ __forceinline__ bool argmax_before(double va, int ia, double vb, int ib) {
    const bool na = va != va, nb = vb != vb;
    if (na || nb) return na && (!nb || ia < ib);
    return va > vb || (va == vb && ia < ib);
}

// One wavefront per frame.
__global__ void __launch_bounds__(256) pitch_pick_kernel(const PitchArgs a) {
    const long long gf = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gf >= (long long)a.C * a.Fc) return;
    const int lane = threadIdx.x & 63;
    const double* st = a.strength + gf * a.Kp;
    double best = st[lane < a.K ? lane : 0];
    int bi = lane < a.K ? lane : 0;
    for (int c = lane + 64; c < a.K; c += 64) {
        const double v = st[c];
        if (argmax_before(v, c, best, bi)) { best = v; bi = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (argmax_before(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    if (lane != 0) return;
    double shift = 0.0;
    if (bi > 0 && bi < a.K - 1) {                               // :392-398, fastParabolicInterp :187-191
        const double y1 = st[bi - 1], y2 = st[bi], y3 = st[bi + 1];
        const double pa = (y1 - 2 * y2 + y3) / 2;
        const double pb = (y3 - y1) / 2;
        shift = -pb / (2 * pa + std::numeric_limits<double>::epsilon());
    }
    // np.interp(idx + shift, arange(L), freqs)   (:402)
    const double xq = (double)bi + shift;
    double f0;
    if (xq != xq) {
        f0 = xq;
    } else if (xq < 0.0) {
        f0 = a.freqs[0];
    } else if (xq >= (double)(a.L - 1)) {
        f0 = a.freqs[a.L - 1];
    } else {
        const int j = (int)floor(xq);
        const double fj = a.freqs[j];
        if ((double)j == xq) {
            f0 = fj;
        } else {
            const double slope = (a.freqs[j + 1] - fj) / ((double)(j + 1) - (double)j);
            f0 = slope * (xq - (double)j) + fj;
        }
    }
    const int chan = (int)(gf / a.Fc);
    const long long f = a.f_start + (gf - (long long)chan * a.Fc);
    a.raw[(0ll * a.C + chan) * a.F + f] = f0;
    a.raw[(1ll * a.C + chan) * a.F + f] = best / 2.56;          // :412
}

// The voiced / unvoiced gate and its carried state (:405-428).  prev = NaN encodes "no previous estimate".
// The previous estimate is either "none" or the raw estimate of the frame before, so the recurrence has one
// bit of state: voiced[f] = level_and_confidence_ok[f] && (!voiced[f-1] || jump_ok[f]), where jump_ok compares
// two raw estimates and does not depend on the state.  One workgroup per channel: every thread folds its
// segment of frames into a map {unvoiced, voiced} -> {unvoiced, voiced}, the maps are chained, and the
// segment is replayed with its true incoming state.
constexpr int kGateThreads = 256;

struct GateFrame {
    bool ok, jump_ok;
    double f0;
};

__device__ __forceinline__ GateFrame gate_frame(const double* raw, int C, long long F, int chan, long long f, double before,
                                                double min_db, double conf, double p_delta) {
    GateFrame g;
    g.f0 = raw[(0ll * C + chan) * F + f];
    const double cf = raw[(1ll * C + chan) * F + f];
    const double db = raw[(2ll * C + chan) * F + f];
    g.ok = !((db < min_db) || (cf < conf));
    g.jump_ok = !(12.0 * fabs(log2(g.f0 / before)) > p_delta);
    return g;
}

__global__ void __launch_bounds__(kGateThreads) pitch_gate_kernel(const double* raw, int C, long long F, double min_db,
                                                                  double conf, double p_delta, double* prev, double* f0_out) {
    __shared__ unsigned char map0[kGateThreads], map1[kGateThreads], incoming[kGateThreads];
    const int chan = blockIdx.x, tid = threadIdx.x;
    const long long seg = (F + kGateThreads - 1) / kGateThreads;
    const long long lo = tid * seg, hi = (lo + seg < F) ? lo + seg : F;
    const double carried = prev[chan];
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const double* f0s = raw + (0ll * C + chan) * F;
    bool m0 = false, m1 = true;       // the segment's map, starting from the identity
    for (long long f = lo; f < hi; ++f) {
        const GateFrame g = gate_frame(raw, C, F, chan, f, f ? f0s[f - 1] : carried, min_db, conf, p_delta);
        const bool o0 = g.ok, o1 = g.ok && g.jump_ok;      // previous frame unvoiced (no jump test, :411-414) / voiced
        m0 = m0 ? o1 : o0;
        m1 = m1 ? o1 : o0;
    }
    map0[tid] = m0;
    map1[tid] = m1;
    __syncthreads();
    if (tid == 0) {
        bool v = carried == carried;
        for (int i = 0; i < kGateThreads; ++i) {
            incoming[i] = v;
            v = v ? map1[i] : map0[i];
        }
    }
    __syncthreads();
    bool v = incoming[tid];
    double last = carried;
    for (long long f = lo; f < hi; ++f) {
        const GateFrame g = gate_frame(raw, C, F, chan, f, f ? f0s[f - 1] : carried, min_db, conf, p_delta);
        v = g.ok && (!v || g.jump_ok);
        last = v ? g.f0 : nan;
        f0_out[(long long)chan * F + f] = last;
    }
    if (hi == F && lo < hi) prev[chan] = last;
}

}  // namespace frt

using namespace frt;

struct frt_pitch {
    int N = 0, hop = 0, C = 0, L = 0, Lp = 0, K = 0, Kp = 0;
    double fs = 0, min_db = 0, conf = 0, p_delta = 0;
    frt_stft* stft = nullptr;
    hipStream_t stream = nullptr;
    size_t scratch_limit = 1ull << 30;
    DeviceBuffer freqs, jidx, kt, lrange, psd, s, strength, raw, prev, eb, stage_in, stage_out;
};

extern "C" void frt_pitch_destroy(frt_pitch* h) {
    if (!h) return;
    if (h->stft) frt_stft_destroy(h->stft);
    DeviceBuffer* bufs[] = {&h->freqs, &h->jidx, &h->kt, &h->lrange, &h->psd, &h->s, &h->strength, &h->raw, &h->prev, &h->eb, &h->stage_in, &h->stage_out};
    for (auto* b : bufs) b->release();
    delete h;
}

extern "C" int frt_pitch_reset(frt_pitch* h) {
    FRT_REQUIRE(h, "frt_pitch_reset: null handle");
    std::vector<double> none(h->C, std::numeric_limits<double>::quiet_NaN());
    FRT_HIP_CHECK(hipMemcpyAsync(h->prev.ptr, none.data(), none.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    return FRT_OK;
}

extern "C" int frt_pitch_set_previous(frt_pitch* h, const double* previous) {
    FRT_REQUIRE(h && previous, "frt_pitch_set_previous: null argument");
    FRT_HIP_CHECK(hipMemcpyAsync(h->prev.ptr, previous, h->C * sizeof(double), hipMemcpyDefault, h->stream));
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    return FRT_OK;
}

extern "C" int frt_pitch_get_previous(frt_pitch* h, double* previous) {
    FRT_REQUIRE(h && previous, "frt_pitch_get_previous: null argument");
    FRT_HIP_CHECK(hipMemcpyAsync(previous, h->prev.ptr, h->C * sizeof(double), hipMemcpyDefault, h->stream));
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    return FRT_OK;
}

extern "C" int frt_pitch_set_scratch_limit(frt_pitch* h, int64_t bytes) {
    FRT_REQUIRE(h && bytes > 0, "frt_pitch_set_scratch_limit: bad argument");
    h->scratch_limit = (size_t)bytes;
    return FRT_OK;
}

extern "C" int frt_pitch_set_gate(frt_pitch* h, double min_db, double conf, double p_delta) {
    FRT_REQUIRE(h, "frt_pitch_set_gate: null handle");
    h->min_db = min_db;
    h->conf = conf;
    h->p_delta = p_delta;
    return FRT_OK;
}

extern "C" int frt_pitch_create(frt_pitch** out, int fft_size, int hop, int n_channels, double sample_rate,
                                const double* log_freqs, int n_log, const double* kernels, int n_candidates,
                                double min_db, double conf, double p_delta) {
    FRT_REQUIRE(out, "frt_pitch_create: null handle pointer");
    *out = nullptr;
    FRT_REQUIRE(log_freqs && kernels, "frt_pitch_create: null table");
    FRT_REQUIRE(n_log >= 2 && n_log <= 4096, "frt_pitch_create: grid of %d frequencies (2..4096 supported)", n_log);
    FRT_REQUIRE(n_candidates >= 1 && n_candidates <= n_log, "frt_pitch_create: %d candidates for a grid of %d", n_candidates, n_log);
    FRT_REQUIRE(sample_rate > 0 && hop >= 1 && n_channels >= 1, "frt_pitch_create: bad sample_rate/hop/n_channels");
    for (int l = 1; l < n_log; ++l)
        FRT_REQUIRE(log_freqs[l] > log_freqs[l - 1], "frt_pitch_create: the frequency grid must increase (index %d)", l);
    frt_pitch* h = new frt_pitch();
    int rc = frt_stft_create(&h->stft, fft_size, hop, n_channels, 64);     // validates fft_size
    if (rc) { delete h; return rc; }
    h->N = fft_size; h->hop = hop; h->C = n_channels; h->fs = sample_rate;
    h->L = n_log; h->Lp = (n_log + 31) / 32 * 32;
    h->K = n_candidates; h->Kp = (n_candidates + kCandPerWave - 1) / kCandPerWave * kCandPerWave;
    h->min_db = min_db; h->conf = conf; h->p_delta = p_delta;
    const int nb = fft_size / 2 + 1;
    const double binw = sample_rate / (double)fft_size;
    // left neighbour of every grid frequency among k * binw, as np.interp's binary search finds it
    std::vector<int> jidx(n_log);
    for (int l = 0; l < n_log; ++l) {
        int j = (int)std::floor(log_freqs[l] / binw);
        if (j < 0) j = 0;
        if (j > nb - 1) j = nb - 1;
        while (j > 0 && (double)j * binw > log_freqs[l]) --j;
        while (j < nb - 1 && (double)(j + 1) * binw <= log_freqs[l]) ++j;
        jidx[l] = j;
    }
    std::vector<double> kt((size_t)(h->Lp + 4) * h->Kp, 0.0);      // rows L .. Lp+3 stay zero (look-ahead of the strength kernel)
    for (int c = 0; c < n_candidates; ++c)
        for (int l = 0; l < n_log; ++l) kt[(size_t)l * h->Kp + c] = kernels[(size_t)c * n_log + l];
    // grid rows [begin, end) in which a block of 256 candidates has any non-zero factor, widened to multiples of 4
    std::vector<int> lrange(2 * (h->Kp / kCandPerWave), 0);
    for (int blk = 0; blk < h->Kp / kCandPerWave; ++blk) {
        int first = -1, last = -1;
        for (int l = 0; l < n_log; ++l) {
            bool any = false;
            for (int c = blk * kCandPerWave; c < (blk + 1) * kCandPerWave && !any; ++c) any = kt[(size_t)l * h->Kp + c] != 0.0;
            if (any) {
                if (first < 0) first = l;
                last = l;
            }
        }
        if (first >= 0) {
            lrange[2 * blk] = first / 4 * 4;
            lrange[2 * blk + 1] = (last + 4) / 4 * 4;          // <= Lp
        }
    }
    std::vector<double> fr(log_freqs, log_freqs + n_log);
    if ((rc = upload(h->freqs, fr)) || (rc = upload(h->jidx, jidx)) || (rc = upload(h->kt, kt)) || (rc = upload(h->lrange, lrange)) ||
        (rc = h->prev.reserve(n_channels * sizeof(double))) || (rc = frt_pitch_reset(h))) {
        frt_pitch_destroy(h);
        return rc;
    }
    *out = h;
    return FRT_OK;
}

extern "C" int frt_pitch_set_stream(frt_pitch* h, void* s) {
    FRT_REQUIRE(h, "frt_pitch_set_stream: null handle");
    h->stream = (hipStream_t)s;
    return frt_stft_set_stream(h->stft, s);
}

extern "C" int64_t frt_pitch_frames_for(const frt_pitch* h, int64_t T) {
    if (!h || T < h->N) return 0;
    return (T - h->N) / h->hop + 1;
}

extern "C" int frt_pitch_track(frt_pitch* h, const double* x, int64_t T, int64_t x_stride, double* f0_out, double* raw_out,
                               int64_t* n_frames_out) {
    FRT_REQUIRE(h, "frt_pitch_track: null handle");
    FRT_REQUIRE(T >= 0 && x_stride >= T, "frt_pitch_track: bad T/x_stride");
    const int64_t F = frt_pitch_frames_for(h, T);
    if (n_frames_out) *n_frames_out = F;
    if (F == 0) return FRT_OK;
    FRT_REQUIRE(x && f0_out, "frt_pitch_track: null buffer");
    const bool dev = is_device_pointer(x);
    FRT_REQUIRE(dev == is_device_pointer(f0_out) && (!raw_out || dev == is_device_pointer(raw_out)),
                "frt_pitch_track: buffers must all be host or all be device memory");
    int rc;
    const size_t out_n = (size_t)h->C * F;
    const double* dx = x;
    double* df0 = f0_out;
    if (!dev) {
        const size_t in_bytes = (size_t)h->C * x_stride * sizeof(double);
        if ((rc = h->stage_in.reserve(in_bytes)) || (rc = h->stage_out.reserve(out_n * sizeof(double)))) return rc;
        FRT_HIP_CHECK(hipMemcpyAsync(h->stage_in.ptr, x, in_bytes, hipMemcpyHostToDevice, h->stream));
        dx = h->stage_in.as<double>();
        df0 = h->stage_out.as<double>();
    }
    if ((rc = h->raw.reserve(3 * out_n * sizeof(double)))) return rc;

    const int nb = h->N / 2 + 1;
    // frames per chunk: scratch (spectra + grid spectra + strengths) bounded by the handle's limit
    const size_t per_frame = ((size_t)nb + h->Lp + h->Kp) * sizeof(double);
    long long fc_max = (long long)(h->scratch_limit / per_frame / h->C);
    fc_max = fc_max / kFramesPerBlock * kFramesPerBlock;
    if (fc_max < kFramesPerBlock) fc_max = kFramesPerBlock;
    const long long fc_alloc = F < fc_max ? F : fc_max;
    const long long padded = ((long long)h->C * fc_alloc + kFramesPerBlock - 1) / kFramesPerBlock * kFramesPerBlock;
    if ((rc = h->psd.reserve((size_t)h->C * fc_alloc * nb * sizeof(double))) ||
        (rc = h->s.reserve(((size_t)padded * h->Lp + 4 * kFramesPerGroup) * sizeof(double))) ||
        (rc = h->strength.reserve((size_t)padded * h->Kp * sizeof(double))))
        return rc;

    PitchArgs a{};
    a.x = dx; a.x_stride = x_stride; a.psd = h->psd.as<double>(); a.s = h->s.as<double>();
    a.strength = h->strength.as<double>(); a.raw = h->raw.as<double>(); a.freqs = h->freqs.as<double>();
    a.jidx = h->jidx.as<int>(); a.kt = h->kt.as<double>();
    a.N = h->N; a.nb = nb; a.hop = h->hop; a.L = h->L; a.Lp = h->Lp; a.K = h->K; a.Kp = h->Kp; a.C = h->C;
    a.F = F; a.binw = h->fs / (double)h->N;
    for (long long f0 = 0; f0 < F; f0 += fc_max) {
        const long long fc = (F - f0) < fc_max ? (F - f0) : fc_max;
        a.f_start = f0; a.Fc = fc;
        int64_t got = 0;
        // spectra of frames [f0, f0 + fc) of every channel: the chunk's slab is [C][fc][nb]
        if ((rc = frt_stft_run(h->stft, FRT_STFT_PSD, dx + f0 * h->hop, h->N + (fc - 1) * h->hop, x_stride, h->psd.ptr, &got))) return rc;
        FRT_REQUIRE(got == fc, "frt_pitch_track: internal frame count mismatch");
        const long long total = (long long)h->C * fc;
        const unsigned groups = (unsigned)((total + kFramesPerGroup - 1) / kFramesPerGroup);
        const unsigned blocks = (unsigned)((total + kFramesPerBlock - 1) / kFramesPerBlock);
        // the strength kernel reads whole 8-frame groups: have the grid kernel fill every group a block touches
        if (h->Lp == 1024 && option(kOptPitchGridTwoPass) <= 0)
            hipLaunchKernelGGL(pitch_loggrid_reg_kernel<32>, dim3(blocks * (kFramesPerBlock / kFramesPerGroup)), dim3(256), 0, h->stream, a);
        else
            hipLaunchKernelGGL(pitch_loggrid_kernel, dim3(blocks * (kFramesPerBlock / kFramesPerGroup)), dim3(256), 0, h->stream, a);
        if (h->N % h->hop == 0 && h->N / h->hop >= 2) {
            const int per_frame = h->N / h->hop;
            const long long n_blocks = fc + per_frame - 1;             // hop-sized blocks the chunk's frames cover
            if ((rc = h->eb.reserve((size_t)h->C * n_blocks * sizeof(double)))) return rc;
            hipLaunchKernelGGL(pitch_block_energy_kernel, dim3((unsigned)((n_blocks * h->C + 3) / 4)), dim3(256), 0, h->stream, dx,
                               (long long)x_stride, h->hop, (long long)f0, n_blocks, h->C, h->eb.as<double>());
            hipLaunchKernelGGL(pitch_level_from_blocks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a,
                               h->eb.as<double>(), n_blocks, per_frame);
        } else {
            hipLaunchKernelGGL(pitch_level_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, h->stream, a);
        }
        hipLaunchKernelGGL(pitch_strength_kernel, dim3(blocks, h->Kp / kCandPerWave), dim3(256), 0, h->stream, a.kt, a.s, a.strength, h->lrange.as<int>(),
                           h->L, h->Lp, h->Kp, total);
        hipLaunchKernelGGL(pitch_pick_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, h->stream, a);
        FRT_HIP_CHECK(hipGetLastError());
        (void)groups;
    }
    hipLaunchKernelGGL(pitch_gate_kernel, dim3(h->C), dim3(kGateThreads), 0, h->stream, a.raw, h->C, (long long)F, h->min_db,
                       h->conf, h->p_delta, h->prev.as<double>(), df0);
    FRT_HIP_CHECK(hipGetLastError());
    if (raw_out) {
        FRT_HIP_CHECK(hipMemcpyAsync(raw_out, h->raw.ptr, 3 * out_n * sizeof(double), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                     h->stream));
    }
    if (!dev) {
        FRT_HIP_CHECK(hipMemcpyAsync(f0_out, df0, out_n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    return FRT_OK;
}
