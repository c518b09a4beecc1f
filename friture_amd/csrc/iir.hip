// iir.hip — K2/K4: exact IIR octave filter bank with decimation, band energies, and the
// stand-alone decimation chain, for gfx950.  Built with -ffp-contract=off: the recurrences replay
// the reference's IEEE double operations one for one.
//
// Reference semantics (paths relative to the reference checkout):
//   direct form II transposed, carried state                       friture/signal/lfilter.py:131-139
//       y[k]  = z[0] + b[0] x[k]
//       z[n]  = z[n+1] + x[k] b[n+1] - y[k] a[n+1]
//       z[-1] = x[k] b[-1] - y[k] a[-1]
//   decimate = 12th-order elliptic low-pass, keep samples 0, 2, 4, ...   friture/signal/decimate.py:27-42
//   octave bank: per octave j the bpo top-octave band-passes (4th order) run on the j-times
//   decimated signal, bands are filled from the top (dec = 2^j)          friture/filter.py:86-118
//   band energy of a block: alpha * sum_i (1-alpha)^(n-1-i) y_i^2 + previous (1-alpha)^n
//                                   friture/signal/exp_smoothing.py:40-56, friture/octavespectrum.py:104
//
// Kernel shape.  A recurrence is serial in time, so the parallel axes are channels, filters, the
// *state index inside a filter*, and — for long batches — time chunks.
//   * A lane group per filter: lane s owns state z[s] and the coefficient pair (b[s+1], a[s+1]).  Per
//     sample the group leader forms y = z[0] + b[0] x, a DPP broadcast hands y to the group, a DPP shift
//     hands z[s+1] to lane s, and every lane updates its state with exactly the reference's operations.
//     The loop-carried dependency is add -> broadcast -> mul -> sub, whatever the filter order.  The
//     12th-order decimator takes a 16-lane DPP row, the 4th-order band-passes a quad: a wavefront carries
//     4 or 16 slots, a slot being one (channel, time chunk, filter) with its own sample stream.
//   * One launch per octave stage and pass.
//   * Sequential mode (one chunk) starts from the carried state and is bit-identical to the
//     reference.  Time-parallel mode needs every chunk's initial state: the zero-state end states of the
//     chunks (a table product, iir_zero_state_kernel), a scan z_{q+1} = A^L z_q + s_q over the chunks
//     (iir_scan_kernel; the powers come from the host), then the output pass from the true initial state of
//     each chunk.  That is the same linear recurrence in a different association order: results agree with
//     the sequential ones to ~1e-10 of the input scale after nine stages (the direct-form decimator
//     amplifies rounding), and the parallelism is C x filters x chunks.
#include <cmath>
#include <mutex>

#include "common.h"
#include "octbank.h"

namespace frt {


#ifndef FRT_IIR_VEC_OUT             // 16-byte stores of the decimated output (measured: no gain over the per-sample stores)
#define FRT_IIR_VEC_OUT 0
#endif

// chunk_end / chunk_init hold, per (channel, filter), nchunks state vectors in a block of nchunks x kStates doubles.  A filter of order
// > 4 uses kStates doubles per chunk; a 4th-order filter packs its four at the front of the block (round 6: a 128-byte line then carries
// four chunks' states instead of one's — the 216-band bank's scans moved 577 MB per call, three quarters of it unused lanes' bytes).
__host__ __device__ constexpr int chunk_state_stride(int order) { return order <= 4 ? 4 : kStates; }

struct IirStageArgs {
    const void* x;             // [C][x_stride] stage input
    long long x_stride;
    int n;                     // samples per channel in this stage
    int in_f32;                // stage-0 input is float
    const double* coef;        // [nfilt][kCoefStride]
    const int* order;          // [nfilt]
    int nfilt;
    int dec_filter;            // index of the decimator among the filters, or -1
    double* state;             // [C][nfilt][kStates] carried state of this stage
    int chunk;                 // samples per chunk (multiple of 64)
    int nchunks;
    int pass;                  // 0 sequential, 1 zero-state scan pass, 2 output pass from chunk_init
    double* chunk_end;         // [C][nfilt][nchunks][kStates] pass 1 result
    const double* chunk_init;  // [C][nfilt][nchunks][kStates] pass 2: every chunk's true initial state (iir_scan_kernel)
    int scan_group;            // chunks per scan row
    int scan_rows;             // scan rows per (channel, filter)
    double* y;                 // band outputs (packed per channel) or null
    long long y_cstride;
    long long y_off[kMaxFilters];   // offset of each filter's band inside a channel's packed row
    double* xnext;             // [C][xnext_stride] decimated output or null
    long long xnext_stride;
    double* eblock;            // [C][nblocks][nbands] zero-state block energies or null
    int eblock_len;            // samples of this stage per energy block (power of two)
    int eblock_shift;          // log2(eblock_len)
    int eblock_mul;            // lane kernel: this stage's energy block spans eblock_mul entries of the block axis (its value goes to the
                               // last of them, zeros to the others): stages whose share of an internal block is under 4 samples
    int nblocks;
    int nbands;
    int band_index[kMaxFilters];    // global band index of each filter (-1 for the decimator)
    const double* alpha;       // [nbands]
    // slot tables, filled by launch_iir_stage
    int n_channels;
    int n_row, n_quad;         // filters that need a 16-lane row (order > 4) / fit a quad (order <= 4)
    int waves_row;             // wavefronts of row slots; quad slots follow
    int dec_lanes;             // lanes per slot of the decimator's mode (16 or 4), 0 without a decimator
    int row_filter[kMaxFilters];
    int quad_filter[kMaxFilters];
    int fused;                      // contracted multiply-adds (energy-only time-parallel calls, see iir_step)
    int vec_x;                      // stage input rows on 16-byte boundaries: vector loads of whole sample groups
    int vec_xnext;                  // likewise the decimated output rows
    int row_energy, quad_energy;    // does any filter of the class feed a band energy?  (the decimator does not: its wavefronts,
                                    // more than half of a 1/3-octave stage, skip the two energy instructions per sample)
    // lane kernel, look-back form (no iir_scan_kernel launch for the stage): a chunk's true initial state is
    // e[q-1] + A^L e[q-2] + ... + (A^L)^(K-1) e[q-K] over the zero-state end states of its K predecessors (chunk_end), e[-1] = the
    // carried state, K = the chunks the filters' decay spans (|A^(L K)| < 1e-20)
    int lookback;                   // K, or 0: initial states from chunk_init
    const double* power_l;          // [nfilt][kStates][kStates] row-major A^L of this stage
    const double* state_in;         // [C][nfilt][kStates] the carried state as it was when the stage began (a snapshot: the last
                                    // chunk's lane replaces `state` while the first chunks' lanes may still be reading)
};

__device__ __forceinline__ double dpp_row_bcast0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150, 0xF, 0xF, false);   // row_newbcast:0
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int T>
__device__ __forceinline__ double dpp_row_bcast(double v) {          // lane T of every 16-lane row, to its row
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane of every row reads a valid lane: with bound_ctrl the "old" operand is dead and costs no move
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + T, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + T, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double dpp_row_shl1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xF, 0xF, true);    // row_shl:1, zero fill
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// DPP moves inside a filter's lane group.  LPS = 16: the group is a DPP row (row_newbcast:0, row_shl:1 with
// zero fill).  LPS = 4: the group is a quad (quad_perm [0,0,0,0] and [1,2,3,3]).
template <int LPS>
__device__ __forceinline__ double group_bcast0(double v) {
    constexpr int ctrl = LPS == 16 ? 0x150 : 0x00;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int LPS>
__device__ __forceinline__ double group_shl1(double v) {
    constexpr int ctrl = LPS == 16 ? 0x101 : 0xF9;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, LPS == 16);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, LPS == 16);
    return __hiloint2double(hi, lo);
}

// One sample of every filter of the wavefront.  The reference's update (lfilter.py:131-139)
//     y = z[0] + b[0] x;   z[s] = z[s+1] + x b[s+1] - y a[s+1];   z[last] = x b[last] - y a[last]
// with z[s+1] arriving by a lane shift.  `keep` is 1 except on the top lane of a quad, where the shift hands the
// lane its own state back: fma(zn, keep, x b) is then exactly x b, and exactly zn + x b (one rounding) elsewhere.
// In a 16-lane row the lanes above the filter order hold zeros, so the top live lane adds 0.
// FUSED (energy-only time-parallel calls, whose contract is the band-energy vector to 1e-5): the same update with its
// multiply-adds contracted — 7 / 8 instead of 9 / 10 VALU instructions per sample in a pass that runs at its issue rate.  The
// band signals then sit ~1e-9 of the input scale from the reference's (the direct-form decimator amplifies the different
// roundings), the energies 1e-9 relative; every path that returns signals keeps the reference's separately rounded operations.
template <int LPS, bool FUSED = false>
__device__ __forceinline__ double iir_step(double x, double b0, double bn, double an, double keep, double& z) {
    if constexpr (FUSED) {
        const double y = group_bcast0<LPS>(__builtin_fma(b0, x, z));
        const double zn = group_shl1<LPS>(z);
        // a 16-lane row shifts in zeros above the order (keep is 1 everywhere); a quad hands the top lane its own state back
        const double t = LPS == 16 ? __builtin_fma(x, bn, zn) : __builtin_fma(zn, keep, x * bn);
        z = __builtin_fma(-y, an, t);
        return y;
    } else {
        const double y = group_bcast0<LPS>(z + b0 * x);
        const double zn = group_shl1<LPS>(z);
        z = __builtin_fma(zn, keep, x * bn) - y * an;
        return y;
    }
}

// `count` consecutive samples, read from and (KEEP) replaced by the outputs in the group's LDS row xy[0..63].
// Straight-line code: no per-sample branch, the energy block boundaries are the caller's.
template <int LPS, bool ENERGY, bool KEEP, bool FUSED = false>
__device__ __forceinline__ void iir_samples(int k0, int count, double* xy, double b0, double bn, double an, double keep,
                                            double& z, double& acc, double decay) {
#pragma unroll 4
    for (int k = k0; k < k0 + count; ++k) {
        const double y = iir_step<LPS, FUSED>(xy[k], b0, bn, an, keep, z);
        if (ENERGY) acc = __builtin_fma(acc, decay, y * y);     // zero-state block energy, Horner form (exp_smoothing.py:40-56); fused: the
                                                                // energies are held to 1e-5, and the pass is instruction-issue bound
        if (KEEP) xy[k] = y;                       // every lane of the group writes the same value to the same slot
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wavefront carries 64 / LPS slots; a slot is one (channel, time chunk, filter) with its own sample stream.
template <int LPS, bool FUSED>
__device__ __forceinline__ void iir_stage_body(const IirStageArgs& a, long long wave, double* lds) {
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x;
    const int sl = lane / LPS, s = lane % LPS;
    const int nf = LPS == 4 ? a.n_quad : a.n_row;
    const int* flist = LPS == 4 ? a.quad_filter : a.row_filter;
    const long long nslots = (long long)a.n_channels * a.nchunks * nf;
    const long long gslot = wave * SPW + sl;
    const bool valid = gslot < nslots;
    // slots past the end shadow the LAST slot (same samples, same arithmetic, nothing stored): a wavefront with spare slots —
    // every streaming call: one or two channels fill 1-2 of the 4 rows — then still runs the straight-line loop below
    // instead of the masked per-sample one (measured on the 2-channel decimator of the delay estimator: 117 -> 18 ns / sample)
    const long long gs = valid ? gslot : nslots - 1;
    const int fi = (int)(gs % nf);
    const long long cq = gs / nf;
    const int q = (int)(cq % a.nchunks), c = (int)(cq / a.nchunks);
    const int f = flist[fi];
    const int ord = a.order[f];
    const bool live = valid && s < ord;
    const bool leader = valid && s == 0;

    const double* cf = a.coef + (size_t)f * kCoefStride;
    const double b0 = cf[0];
    const double bn = live ? cf[s + 1] : 0.0;
    const double an = live ? cf[kMaxOrder + 1 + s + 1] : 0.0;
    const double keep = (LPS == 4 && s == 3) ? 0.0 : 1.0;

    const size_t sidx = ((size_t)c * a.nfilt + f) * kStates + s;
    const size_t cidx = ((size_t)c * a.nfilt + f) * a.nchunks * kStates + (size_t)q * (LPS == 4 ? 4 : kStates) + s;      // (quad slots: order <= 4)
    double z = 0.0;
    if (live) {
        if (a.pass == 0) z = a.state[sidx];
        else if (a.pass == 2) {
            z = a.chunk_init[cidx];      // the chunk's true initial state (iir_scan_kernel)
        }
    }

    const int band = a.band_index[f];
    const bool energy = a.eblock != nullptr && a.pass != 1 && (LPS == 16 ? a.row_energy : a.quad_energy) != 0;      // uniform
    const bool my_energy = energy && band >= 0 && leader;
    const double alpha = (energy && band >= 0) ? a.alpha[band] : 0.0;
    const double decay = 1.0 - alpha;
    double acc = 0.0;
    double* eout = a.eblock + (size_t)c * a.nblocks * a.nbands + (band >= 0 ? band : 0);
    const int elen = a.eblock_len;

    const bool write_y = a.pass != 1 && a.y != nullptr;
    const bool write_dec = a.pass != 1 && a.xnext != nullptr;
    // uniform: the band signals when asked for, the decimator's output in the wavefronts that carry the decimator
    const bool keep_out = write_y || (write_dec && a.dec_lanes == LPS);
    const bool is_dec = f == a.dec_filter;

    const long long start = (long long)q * a.chunk;
    long long stop = start + a.chunk;
    if (stop > a.n) stop = a.n;
    double* xy = lds + sl * 64;
    const long long xrow = (long long)c * a.x_stride;

    for (int g0 = 0; g0 < a.chunk; g0 += 64) {                        // a.chunk is a multiple of 64
        const long long base = start + g0;
        const long long left = stop - base;
        const int cnt = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
        // the LPS lanes of a slot fetch its 64 samples: 16-byte loads for a whole group on aligned rows (a quarter / half of
        // the load, convert and LDS-write instructions — the pass runs at its VALU issue rate, and this prologue was 4 of
        // its 15.6 instructions per sample), one sample at a time for the ragged end of a channel
        if (a.vec_x && cnt == 64) {
            // (every load of the group first, then the conversions and LDS writes: written load-by-load the compiler gave all of
            // them one register and waited behind each — eight round trips per group of a quad slot's float64 samples)
            if (a.in_f32) {
                const float* xp = (const float*)a.x + xrow + base;
                float4 v[SPW / 4];
#pragma unroll
                for (int j = 0; j < SPW / 4; ++j) v[j] = *(const float4*)(xp + 4 * (s + LPS * j));
#pragma unroll
                for (int j = 0; j < SPW / 4; ++j) {
                    const int k = 4 * (s + LPS * j);
                    *(double2*)(xy + k) = double2{(double)v[j].x, (double)v[j].y};
                    *(double2*)(xy + k + 2) = double2{(double)v[j].z, (double)v[j].w};
                }
            } else {
                const double* xp = (const double*)a.x + xrow + base;
                double2 v[SPW / 2];
#pragma unroll
                for (int j = 0; j < SPW / 2; ++j) v[j] = *(const double2*)(xp + 2 * (s + LPS * j));
#pragma unroll
                for (int j = 0; j < SPW / 2; ++j) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y));      // (all of them requested before the first is stored)
#pragma unroll
                for (int j = 0; j < SPW / 2; ++j) *(double2*)(xy + 2 * (s + LPS * j)) = v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < SPW; ++j) {
                const int k = s + LPS * j;
                double v = 0.0;
                if (k < cnt) v = a.in_f32 ? (double)((const float*)a.x)[xrow + base + k] : ((const double*)a.x)[xrow + base + k];
                xy[k] = v;
            }
        }
        wave_lds_sync();
        if (__all(cnt == 64)) {
            if (!energy) {
                if (keep_out) iir_samples<LPS, false, true, FUSED>(0, 64, xy, b0, bn, an, keep, z, acc, decay);
                else iir_samples<LPS, false, false, FUSED>(0, 64, xy, b0, bn, an, keep, z, acc, decay);
            } else if (elen >= 64) {
                // energy blocks are whole multiples of a group: boundaries only between groups (chunks start on
                // block boundaries, so the phase g0 mod elen is the same for every slot)
                if ((g0 & (elen - 1)) == 0) acc = 0.0;
                if (keep_out) iir_samples<LPS, true, true, FUSED>(0, 64, xy, b0, bn, an, keep, z, acc, decay);
                else iir_samples<LPS, true, false, FUSED>(0, 64, xy, b0, bn, an, keep, z, acc, decay);
                if (my_energy && ((g0 + 64) & (elen - 1)) == 0) eout[(size_t)(base >> a.eblock_shift) * a.nbands] = alpha * acc;
            } else {
                for (int k0 = 0; k0 < 64; k0 += elen) {               // short blocks of the low-rate stages
                    acc = 0.0;
                    if (keep_out) iir_samples<LPS, true, true, FUSED>(k0, elen, xy, b0, bn, an, keep, z, acc, decay);
                    else iir_samples<LPS, true, false, FUSED>(k0, elen, xy, b0, bn, an, keep, z, acc, decay);
                    if (my_energy) eout[(size_t)((base + k0) >> a.eblock_shift) * a.nbands] = alpha * acc;
                }
            }
        } else {
            // ragged end of a channel (or slots past the end): same arithmetic, state frozen beyond the slot's count
            for (int k = 0; k < 64; ++k) {
                if (!__any(k < cnt)) break;
                const bool on = k < cnt;
                double zt = z;
                const double y = iir_step<LPS, FUSED>(xy[k], b0, bn, an, keep, zt);
                z = on ? zt : z;
                if (energy) {
                    if (((g0 + k) & (elen - 1)) == 0) acc = 0.0;
                    const double at = __builtin_fma(acc, decay, y * y);
                    acc = on ? at : acc;
                    if (my_energy && on && ((g0 + k + 1) & (elen - 1)) == 0)
                        eout[(size_t)((base + k) >> a.eblock_shift) * a.nbands] = alpha * acc;
                }
                if (keep_out) xy[k] = y;
            }
        }
        if (keep_out) {
            wave_lds_sync();
            if (is_dec) {
                if (write_dec) {
                    double* xn = a.xnext + (long long)c * a.xnext_stride + (base >> 1);       // base is a multiple of 64
                    if (FRT_IIR_VEC_OUT && a.vec_xnext && cnt == 64) {
                        // a whole group: the 32 even samples as 16-byte stores of two, dealt to the slot's lanes
#pragma unroll
                        for (int j = 0; j < (SPW + 3) / 4; ++j) {
                            const int m = 2 * (s + LPS * j);                                      // decimated index of the pair
                            if (m < 32 && valid) *(double2*)(xn + m) = double2{xy[2 * m], xy[2 * m + 2]};
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < SPW; ++j) {
                            const int k = s + LPS * j;
                            if (!(k & 1) && k < cnt && valid) xn[k >> 1] = xy[k];
                        }
                    }
                }
            } else if (write_y) {
                double* yrow = a.y + (long long)c * a.y_cstride + a.y_off[f] + base;
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    const int k = s + LPS * j;
                    if (k < cnt && valid) yrow[k] = xy[k];
                }
            }
        }
        wave_lds_sync();
    }

    if (live) {
        if (a.pass == 1) a.chunk_end[cidx] = z;
        else if (q == a.nchunks - 1) a.state[sidx] = z;
    }
}

// grid.x = waves of 16-lane-row slots (filters of order > 4: the decimator) followed by waves of quad slots.
__global__ void __launch_bounds__(64) iir_stage_kernel(const IirStageArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[16 * 64];
    if (a.fused) {
        if ((int)blockIdx.x < a.waves_row) iir_stage_body<16, true>(a, blockIdx.x, lds);
        else iir_stage_body<4, true>(a, (long long)blockIdx.x - a.waves_row, lds);
    } else {
        if ((int)blockIdx.x < a.waves_row) iir_stage_body<16, false>(a, blockIdx.x, lds);
        else iir_stage_body<4, false>(a, (long long)blockIdx.x - a.waves_row, lds);
    }
}

// Fills the slot tables of a stage from the filter orders and launches it.
static int launch_iir_stage(IirStageArgs a, const int* orders, int n_channels, hipStream_t stream) {
    a.n_channels = n_channels;
    a.n_row = a.n_quad = 0;
    a.dec_lanes = 0;
    a.vec_x = ((uintptr_t)a.x % 16 == 0) && (a.x_stride % (a.in_f32 ? 4 : 2) == 0);
    a.vec_xnext = a.xnext && ((uintptr_t)a.xnext % 16 == 0) && (a.xnext_stride % 2 == 0);
    a.row_energy = a.quad_energy = 0;
    for (int f = 0; f < a.nfilt; ++f) {
        if (orders[f] > 4) {
            a.row_filter[a.n_row++] = f;
            a.row_energy |= a.band_index[f] >= 0;
        } else {
            a.quad_filter[a.n_quad++] = f;
            a.quad_energy |= a.band_index[f] >= 0;
        }
        if (f == a.dec_filter) a.dec_lanes = orders[f] > 4 ? 16 : 4;
    }
    const long long per = (long long)n_channels * a.nchunks;
    const long long wr = (per * a.n_row + 3) / 4, wq = (per * a.n_quad + 15) / 16;
    FRT_REQUIRE(wr + wq < (1ll << 31), "iir stage: too many wavefronts");
    a.waves_row = (int)wr;
    if (wr + wq == 0) return FRT_OK;
    hipLaunchKernelGGL(iir_stage_kernel, dim3((unsigned)(wr + wq)), dim3(64), 0, stream, a);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

// ---- time-parallel mode, output pass of ENERGY-ONLY calls: one LANE per (channel, time chunk) ------------------------
// The slot kernel above spreads ONE filter over the lanes of a quad or a DPP row (lane = state index): right for the
// sequential mode, where a channel offers a handful of independent recurrences, but a wavefront then advances 16 band
// filters (or 4 decimators) by one sample in ~10 instructions.  A time-parallel stage has thousands of chunks per
// channel, each an independent recurrence once its initial state is known: here a lane owns one chunk, keeps the whole
// state vector of its filters in registers and steps through the chunk's samples with plain multiply-adds — the
// coefficients are wave-uniform (every lane runs the same filters), so they sit in scalar registers and the loop is
// 2 ORD + 1 (+ 2 for the block energy) float64 instructions per sample and filter for 64 chunks at once: 11 per band
// filter instead of 10 per 16 filter-steps, i.e. ~5.8x the filter-steps per instruction (the decimator: 25 instead of 7 per
// 4).  Multiply-adds are contracted like the slot kernel's FUSED form (energies to 1e-5; band signals are never
// produced here: calls that return signals keep the slot kernel and the reference's separately rounded operations).
// A wavefront = 64 consecutive chunks of one channel x one filter group (up to kLaneBands band filters, or the decimator,
// which also writes the even samples of its output as the next stage's input).
#ifndef FRT_LANE_BANDS
#define FRT_LANE_BANDS 3
#endif
#ifndef FRT_LANE_PREFETCH
#define FRT_LANE_PREFETCH 1
#endif
constexpr int kLanePrefetch = FRT_LANE_PREFETCH;      // trips of 16 samples a lane's requests run ahead; measured 1 / 2 / 4: equal / equal / slower at 216 bands (registers)
static_assert(kLanePrefetch == 1 || kLanePrefetch == 2 || kLanePrefetch == 4, "prefetch ring of 1, 2 or 4 trips");
constexpr int kLaneBands = FRT_LANE_BANDS;      // measured: one band filter per wavefront re-reads the samples per filter and is 15-70 % slower (bpo 3 / 24)

// (bx, lane): the wavefront's index along the launch's first axis — channel x 64-chunk group — and the lane inside it
// LPC (lanes per chunk) > 1: the band filters of a group on LPC neighbouring lanes of ONE chunk, a filter per lane (NF = 1; `nlive`
// of them exist), 64 / LPC chunks per wavefront — a third of the dependent float64 instructions per sample and wave and three
// times the wavefronts of the filter-group-per-lane form (iir_lane_split_kernel).  The lanes of a chunk read the same samples.
// STAGED (round 6, iir_lane_col_kernel): the samples come from a tile in LDS that the wavefronts of the chunk column — every filter
// group of the same 64 chunks, one workgroup — fill together with coalesced loads, instead of from requests in which every lane
// walks its own cache lines (64 lines per load instruction, again per filter group).
#ifndef FRT_COL_F64_PIECES
#define FRT_COL_F64_PIECES 16
#endif
constexpr int kColRows = 64;                                      // a tile: 64 chunks (rows) x PIECES pieces of 16 bytes
template <bool F32> struct ColTile {
    static constexpr int kPieces = F32 ? 16 : FRT_COL_F64_PIECES; // 64 floats; 32 or 64 doubles
    static constexpr int kRowBytes = kPieces * 16;                // rows are contiguous: an LDS-DMA instruction (1 KB) fills 1024 / kRowBytes of them
    static constexpr int kBytes = kColRows * kRowBytes;
    static constexpr int kSamples = kRowBytes / (F32 ? 4 : 8);
};
// piece j of row r sits in slot j ^ (r & 15) of the row: sixteen lanes reading piece j of sixteen consecutive rows hit sixteen
// different 16-byte bank groups (unswizzled, rows 256 or 512 bytes apart, they would all hit the same one)

template <int NF, int ORD, bool DEC, bool F32, int LPC = 1, bool STAGED = false>
__device__ __forceinline__ void iir_lane_body(const IirStageArgs& a, int f0_group, int bx, int lane, int nlive = NF, char* lds = nullptr,
                                              int tid_col = 0, int nt_col = 64, bool live = true) {
    static_assert(LPC == 1 || (NF == 1 && !DEC), "a filter per lane");
    static_assert(!STAGED || LPC == 1, "staged tiles hold a chunk per lane");
    constexpr int CPW = 64 / LPC;                               // chunks per wavefront
    const int cl = LPC == 1 ? lane : (lane * 43) >> 7;          // lane / 3 for lane < 64 (LPC is 1 or 3)
    static_assert(LPC == 1 || LPC == 3, "lane / LPC by multiply-shift");
    const int ml = lane - cl * LPC;                             // the lane's filter inside the group
    const int nblk = (a.nchunks + CPW - 1) / CPW;
    const int c = bx / nblk;
    const int q = (bx - c * nblk) * CPW + cl;
    const bool valid = q < a.nchunks && cl < CPW && ml < nlive && live;
    const int qc = (q < a.nchunks && cl < CPW) ? q : a.nchunks - 1;      // lanes past the end shadow the last chunk, store nothing
    const int f0 = f0_group + (LPC > 1 && ml < nlive ? ml : 0);          // (per lane when LPC > 1)
    const int L = a.chunk;

    // initial state of every filter of the group: formed by iir_scan_kernel (zero-start prefix + (A^L)^i x the scan row's true
    // start).  Until round 4 every lane composed it here from three tables — per-lane gathers of a 16 x 16 power, 13-15 dependent
    // round trips, most of what a low-rate stage's launch lasted.
    static_assert(ORD % 2 == 0, "states are read as pairs of doubles");
    double z[NF][ORD], acc[NF], alpha[NF], decay[NF];
    int bandv[NF];
    typedef const double __attribute__((address_space(4))) * ktable;
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        const int f = f0 + m;
        if (LPC == 1 && a.lookback > 0) {
            // Round 6: the look-back form — no scan launch in front of this one (19 / 15 us of the two highest-rate stages, nearly
            // independent of their size).  Horner over the K predecessors' zero-state end states: K - 1 products with the wave-uniform
            // A^L (scalar loads), a lane's end states 128 contiguous bytes each; ~1-2 % of the chunk's own arithmetic at K = 6 .. 12.
            const double* ends = a.chunk_end + ((size_t)c * a.nfilt + f) * a.nchunks * kStates;
            const double* carried = a.state_in + ((size_t)c * a.nfilt + f) * kStates;
            unsigned long long pw_at = (unsigned long long)(uintptr_t)(a.power_l + (size_t)f * kStates * kStates);
            // e[i]: chunk i's zero-state end state; the carried state for i = -1; zero in front of that.  The loads are unconditional (a
            // lane in front of the channel's first chunk reads the carried state and zeroes it when it USES it) and the next step's
            // request is pinned in front of this step's product: behind `if (k > 1)` and a per-lane `i >= -1 ? load : 0` every request was
            // awaited where it was issued (the merge copies again) — a memory round trip per Horner step, most of the 1.4-1.9 us a
            // step cost (192 multiply-adds are ~0.5 us).
            auto request_end = [&](int i, double (&e)[ORD]) {
                const double* src = i >= 0 ? ends + (size_t)i * chunk_state_stride(ORD) : carried;
#pragma unroll
                for (int t = 0; t < ORD; t += 2) {
                    const double2 v = *(const double2*)(src + t);
                    e[t] = v.x;
                    e[t + 1] = v.y;
                }
            };
            double zz[ORD], en[ORD];
            request_end(qc - a.lookback, zz);
            request_end(qc - a.lookback + (a.lookback > 1 ? 1 : 0), en);
            {
                const bool none = qc - a.lookback < -1;
#pragma unroll
                for (int t = 0; t < ORD; ++t) zz[t] = none ? 0.0 : zz[t];
            }
            for (int k = a.lookback - 1; k >= 1; --k) {
                double e[ORD];
                const bool none = qc - k < -1;
#pragma unroll
                for (int t = 0; t < ORD; ++t) {
                    e[t] = none ? 0.0 : en[t];
                    asm volatile("" : "+v"(e[t]));
                }
                request_end(qc - (k > 1 ? k : 2) + 1, en);        // the next step's end state travels during this product (the last step re-reads)
                asm volatile("" ::: "memory");
                // (the table's address as a value made in this trip: hoisted out of the loop its ORD x ORD entries — 288 scalar
                // registers for the decimator — are spilled into vector-register lanes, the pass's coefficients with them)
                asm volatile("" : "+s"(pw_at));
                const ktable pw = (ktable)pw_at;
                // column by column: ORD independent accumulations per column (row by row each row is one dependent chain of ORD
                // multiply-adds, and a lone wavefront waits out every one of them)
#pragma unroll
                for (int t = 0; t < ORD; ++t)
#pragma unroll
                    for (int r = 0; r < ORD; ++r) e[r] = __builtin_fma(pw[r * kStates + t], zz[t], e[r]);
#pragma unroll
                for (int t = 0; t < ORD; ++t) zz[t] = e[t];
            }
#pragma unroll
            for (int t = 0; t < ORD; ++t) z[m][t] = zz[t];
        } else {
            const double* init = a.chunk_init + ((size_t)c * a.nfilt + f) * a.nchunks * kStates + (size_t)qc * chunk_state_stride(ORD);
#pragma unroll
            for (int t = 0; t < ORD; t += 2) {
                const double2 iv = *(const double2*)(init + t);
                z[m][t] = iv.x;
                z[m][t + 1] = iv.y;
            }
        }
        // (a stage's band filters carry consecutive band indices: a lane's own is the group's first plus its offset, no
        // lane-indexed read of the argument block)
        const int band = LPC > 1 ? a.band_index[f0_group] + (f - f0_group) : a.band_index[f];
        bandv[m] = band;
        alpha[m] = (!DEC && band >= 0) ? a.alpha[band] : 0.0;
        decay[m] = 1.0 - alpha[m];
        acc[m] = 0.0;
    }
    // coefficients: wave-uniform, read ONCE through the constant address space (s_load: the table is never written by a
    // kernel; as plain global loads the compiler re-reads all of them every iteration, next to stores it cannot prove
    // disjoint) and used as scalar operands of the multiply-adds
    double cb[NF][ORD + 1], ca[NF][ORD + 1];
#pragma unroll
    for (int m = 0; m < NF; ++m) {
        const ktable t = (ktable)(uintptr_t)(a.coef + (size_t)(f0 + m) * kCoefStride);
#pragma unroll
        for (int i = 0; i <= ORD; ++i) {
            cb[m][i] = t[i];
            ca[m][i] = i ? -t[kMaxOrder + 1 + i] : 0.0;          // the update subtracts a[i] y: the negated value is what fma takes
        }
    }

    const long long s0 = (long long)qc * L;                    // first sample of the chunk in its channel's stage signal
    const float* xf = (const float*)a.x + (long long)c * a.x_stride + s0;
    const double* xd = (const double*)a.x + (long long)c * a.x_stride + s0;
    double* xn = DEC && a.xnext ? a.xnext + (long long)c * a.xnext_stride + (s0 >> 1) : nullptr;
    const int elen = a.eblock_len;
    const bool energy = !DEC && a.eblock != nullptr;
    const bool big_blocks = !energy || elen >= 16;              // (uniform)
    constexpr bool f32 = F32;

    // Sixteen samples per trip (chunks are multiples of 64) — one 64-byte (float) or 128-byte (double) piece of the lane's own
    // stream — requested kLanePrefetch trips before they are filtered and converted only when their turn comes.  Every lane walks
    // its own cache lines (the chunks lie L samples apart), so a request is an L2 round trip, often an HBM one, and a wavefront is
    // alone on its SIMD at the high-rate stages: with four samples per trip every trip began with a wait (round 3); one trip ahead
    // (16 samples = ~2200 cycles of arithmetic) covers it — two or four trips ahead measure equal (round 4), the pass is not waiting
    // for memory.
    constexpr int G = 16, D = kLanePrefetch;
    static_assert(64 % (G * D) == 0, "a chunk is whole rounds of the prefetch ring");
    float4 rf[D][4];
    double2 rd[D][8];
    double ydp[G / 2] = {};
    typedef ColTile<F32> Tile;
    constexpr int S = Tile::kSamples, ESZ = F32 ? 4 : 8;       // (staged) samples of a tile row, bytes of a sample
    const char* tile_row = lds + lane * Tile::kRowBytes;        // (staged) this lane's row of the tile being filtered
    constexpr int NI = F32 ? 4 : 8;                              // (staged) 16-byte pieces of a trip
    int soff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) soff[i] = (i ^ (lane & (NI - 1))) << 4;
    const int sbase = (lane & 15 & ~(NI - 1)) << 4;
    auto request = [&](auto slot, int k) {
        constexpr int d = decltype(slot)::value;
        if constexpr (STAGED) {
            // piece pj + i (pj a multiple of NI) sits in slot (pj + i) ^ sw = (pj ^ (sw & ~(NI - 1))) | (i ^ (sw & (NI - 1))): the second
            // term is the lane's own for the whole pass (soff), the first one XOR per trip
            const char* src = tile_row + ((((k & (S - 1)) * ESZ)) ^ sbase);
            if (f32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rf[d][i] = *(const float4*)(src + soff[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) rd[d][i] = *(const double2*)(src + soff[i]);
            }
        } else if (f32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rf[d][i] = *(const float4*)(xf + k + 4 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) rd[d][i] = *(const double2*)(xd + k + 2 * i);
        }
    };
    // bigc: the energy blocks are whole trips (>= 16 samples: every stage but the lowest rates) — a block begins and ends only where a
    // trip does, and the per-group tests (eight scalar compare-and-branch pairs per trip, issue slots of a lone wavefront) are gone
    auto trip = [&](auto slot, auto bigc, int k0) {
        constexpr int d = decltype(slot)::value;
        constexpr bool BIG = decltype(bigc)::value;
        double xg[G];
        if constexpr (STAGED) request(slot, k0);                 // an LDS read: ~100 cycles in front of ~3000 of arithmetic
        if (f32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { xg[4 * i] = rf[d][i].x; xg[4 * i + 1] = rf[d][i].y; xg[4 * i + 2] = rf[d][i].z; xg[4 * i + 3] = rf[d][i].w; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { xg[2 * i] = rd[d][i].x; xg[2 * i + 1] = rd[d][i].y; }
        }
        // The next piece is requested on EVERY trip (the last one re-reads its own: the address stays inside the chunk) and only after
        // this trip's samples exist as doubles.  Round 6: behind `if (k0 + G D < L)` the ring's registers were a merge of "loaded" and
        // "kept", the compiler put the merge's copies into the conditional block right behind the loads, and every trip of every lane
        // pass began with a wait for the memory round trip it had just started — the prefetch never ran ahead (which is also why one,
        // two and four trips of distance measured equal in round 4).
#pragma unroll
        for (int i = 0; i < G; ++i) asm volatile("" : "+v"(xg[i]));
        // the decimator's outputs of the PREVIOUS trip leave here, in front of the request: stored at the end of their own trip, the
        // next trip's wait for its samples (one counter for loads and stores) would wait for the write acknowledgements too
        if (DEC && xn && valid && k0 > 0) {
#pragma unroll
            for (int i = 0; i < G / 4; ++i) *(double2*)(xn + ((k0 - G) >> 1) + 2 * i) = double2{ydp[2 * i], ydp[2 * i + 1]};
        }
        if constexpr (!STAGED) {
            request(slot, k0 + G * D < L ? k0 + G * D : k0);
            asm volatile("" ::: "memory");      // (nothing but arithmetic follows in the trip: without this the request moves to the top of the NEXT trip)
        }
        double yd[G / 2];
#pragma unroll
        for (int u4 = 0; u4 < G; u4 += 4) {
            const int k = k0 + u4;
            if (energy && (BIG ? (u4 == 0 && (k0 & (elen - 1)) == 0) : (k & (elen - 1)) == 0)) {
#pragma unroll
                for (int m = 0; m < NF; ++m) acc[m] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double x = xg[u4 + u];
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    const double y = __builtin_fma(cb[m][0], x, z[m][0]);
#pragma unroll
                    for (int st = 0; st + 1 < ORD; ++st) z[m][st] = __builtin_fma(ca[m][st + 1], y, __builtin_fma(cb[m][st + 1], x, z[m][st + 1]));
                    z[m][ORD - 1] = __builtin_fma(ca[m][ORD], y, cb[m][ORD] * x);
                    if (!DEC) acc[m] = __builtin_fma(acc[m], decay[m], y * y);
                    else if (!(u & 1)) yd[(u4 + u) / 2] = y;                  // decimate.py:41: samples 0, 2, 4, ...
                }
            }
            if (energy && (BIG ? (u4 == G - 4 && ((k0 + G) & (elen - 1)) == 0) : ((k + 4) & (elen - 1)) == 0) && valid) {
                const long long blk = ((s0 + k) >> a.eblock_shift) * a.eblock_mul;      // first entry of the block axis this block spans
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    const int band = bandv[m];
                    if (band >= 0) {
                        double* e = a.eblock + ((size_t)c * a.nblocks + blk) * a.nbands + band;
                        for (int i = 0; i + 1 < a.eblock_mul; ++i) e[(size_t)i * a.nbands] = 0.0;
                        e[(size_t)(a.eblock_mul - 1) * a.nbands] = alpha[m] * acc[m];
                    }
                }
            }
        }
        if (DEC) {
#pragma unroll
            for (int i = 0; i < G / 2; ++i) ydp[i] = yd[i];
        }
    };
    if constexpr (STAGED) {
        // The NEXT tile is copied while this one is filtered, global memory -> LDS without registers (global_load_lds_dwordx4: 64 lanes x
        // 16 bytes land contiguously at M0): an instruction fills four rows, lane = 16 x (row of the four) + slot, and fetches piece
        // slot ^ (row & 15) of its row — sixteen lanes read one chunk's 256 contiguous bytes, 8 cache lines per instruction instead
        // of 64.  The column's wavefronts (wave_col of nw_col) deal the tile's sixteen instructions round-robin.  Issued from inline
        // assembly (the compiler's wait counting does not see them): vmcnt(0) by hand in front of the barrier that ends a tile.
        const int q0 = (bx - c * nblk) * CPW;
        const char* xcol = (const char*)a.x + (long long)c * a.x_stride * ESZ;
        const uint32_t lds_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
        const int wave_col = __builtin_amdgcn_readfirstlane(tid_col >> 6), nw_col = __builtin_amdgcn_readfirstlane(nt_col >> 6);
        auto fetch = [&](int t0, int buf) {
            constexpr int LPR = Tile::kPieces, RPI = 64 / LPR;   // lanes per row, rows per instruction
            for (int n = wave_col; n < kColRows / RPI; n += nw_col) {
                const int row = RPI * n + lane / LPR, slot = lane % LPR;
                const int qq = q0 + row < a.nchunks ? q0 + row : a.nchunks - 1;          // rows past the end shadow the last chunk
                const char* src = xcol + ((long long)qq * L + t0) * ESZ + ((slot ^ (row & 15)) << 4);
                const uint32_t dst = lds_at + buf * Tile::kBytes + n * 1024;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory", "m0");
            }
        };
        fetch(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
        __syncthreads();
        int buf = 0;
        for (int t0 = 0; t0 < L; t0 += S, buf ^= 1) {
            if (t0 + S < L) fetch(t0 + S, buf ^ 1);
            tile_row = lds + buf * Tile::kBytes + lane * Tile::kRowBytes;
            if (big_blocks) for (int k0 = t0; k0 < t0 + S; k0 += G) trip(std::integral_constant<int, 0>{}, std::true_type{}, k0);
            else for (int k0 = t0; k0 < t0 + S; k0 += G) trip(std::integral_constant<int, 0>{}, std::false_type{}, k0);
            __builtin_amdgcn_s_waitcnt(0x0F70);                 // the next tile has landed (and this wavefront's stores have left)
            __syncthreads();
        }
    } else {
        request(std::integral_constant<int, 0>{}, 0);
        if constexpr (D > 1) request(std::integral_constant<int, 1>{}, G);
        if constexpr (D > 2) {
            request(std::integral_constant<int, 2>{}, 2 * G);
            request(std::integral_constant<int, 3>{}, 3 * G);
        }
        auto walk = [&](auto bigc) {
            for (int k0 = 0; k0 < L; k0 += G * D) {
                trip(std::integral_constant<int, 0>{}, bigc, k0);
                if constexpr (D > 1) trip(std::integral_constant<int, 1>{}, bigc, k0 + G);
                if constexpr (D > 2) {
                    trip(std::integral_constant<int, 2>{}, bigc, k0 + 2 * G);
                    trip(std::integral_constant<int, 3>{}, bigc, k0 + 3 * G);
                }
            }
        };
        if (big_blocks) walk(std::true_type{});
        else walk(std::false_type{});
    }
    if (DEC && xn && valid) {
#pragma unroll
        for (int i = 0; i < G / 4; ++i) *(double2*)(xn + ((L - G) >> 1) + 2 * i) = double2{ydp[2 * i], ydp[2 * i + 1]};
    }
    if (valid && q == a.nchunks - 1) {                          // the stage's carried state: end of the channel's last chunk
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int st = 0; st < ORD; ++st) a.state[((size_t)c * a.nfilt + f0 + m) * kStates + st] = z[m][st];
    }
}

// The decimator of a chunk on a PAIR of lanes: lane A (even) owns states 0..5, lane B (odd) states 6..11 of the 12th-order DF2T
// recurrence  y = b0 x + z0;  z_s <- b_{s+1} x - a_{s+1} y + z_{s+1}  (lfilter.py:131-139, re-associated like the rest of the
// time-parallel mode).  Per sample: A forms y, a quad permute hands it to B, another hands B's z6 (its old value) to A for z5;
// then both lanes update their six states — 13 multiply-adds + 5 moves per lane instead of 25 multiply-adds on one, and the
// dependent chain from a sample's y to the next is  y -> permute -> z0 -> y  whatever the order.  32 chunks per wavefront.
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <bool F32>
__device__ __forceinline__ void iir_lane_dec_pair_body(const IirStageArgs& a, int bx, int lane) {
    constexpr int H = 6, CPW = 32;
    const int f = a.dec_filter;
    const int half = lane & 1, cl = lane >> 1;
    const int nblk = (a.nchunks + CPW - 1) / CPW;
    const int c = bx / nblk;
    const int q = (bx - c * nblk) * CPW + cl;
    const bool valid = q < a.nchunks;
    const int qc = valid ? q : a.nchunks - 1;
    const int L = a.chunk;
    double u[H];
    {
        const double* init = a.chunk_init + (((size_t)c * a.nfilt + f) * a.nchunks + qc) * kStates + H * half;
#pragma unroll
        for (int t = 0; t < H; t += 2) {
            const double2 iv = *(const double2*)(init + t);
            u[t] = iv.x;
            u[t + 1] = iv.y;
        }
    }
    // this lane's coefficients: b[H half + s + 1], -a[H half + s + 1] for its states s = 0..5, and b0 (lane A forms y)
    // (through the constant address space like the band lanes' tables: plain global loads would be re-read behind every store)
    typedef const double __attribute__((address_space(4))) * ktable;
    const ktable tab = (ktable)(uintptr_t)(a.coef + (size_t)f * kCoefStride);
    const double b0 = tab[0];
    double cb[H], ca[H];
#pragma unroll
    for (int sidx = 0; sidx < H; ++sidx) {
        cb[sidx] = tab[H * half + sidx + 1];
        ca[sidx] = -tab[kMaxOrder + 1 + H * half + sidx + 1];
    }
    const bool is_a = half == 0;
    const long long s0 = (long long)qc * L;
    const float* xf = (const float*)a.x + (long long)c * a.x_stride + s0;
    const double* xd = (const double*)a.x + (long long)c * a.x_stride + s0;
    double* xn = a.xnext ? a.xnext + (long long)c * a.xnext_stride + (s0 >> 1) : nullptr;
    constexpr int G = 16;
    float4 rf[4];
    double2 rd[8];
    auto request = [&](int k) {
        if (F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rf[i] = *(const float4*)(xf + k + 4 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) rd[i] = *(const double2*)(xd + k + 2 * i);
        }
    };
    request(0);
    double ydp[G / 2] = {};
    for (int k0 = 0; k0 < L; k0 += G) {
        double xg[G];
        if (F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { xg[4 * i] = rf[i].x; xg[4 * i + 1] = rf[i].y; xg[4 * i + 2] = rf[i].z; xg[4 * i + 3] = rf[i].w; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { xg[2 * i] = rd[i].x; xg[2 * i + 1] = rd[i].y; }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) asm volatile("" : "+v"(xg[i]));
        if (xn && valid && is_a && k0 > 0) {              // the previous trip's outputs, in front of the request: see iir_lane_body
#pragma unroll
            for (int i = 0; i < G / 4; ++i) *(double2*)(xn + ((k0 - G) >> 1) + 2 * i) = double2{ydp[2 * i], ydp[2 * i + 1]};
        }
        request(k0 + G < L ? k0 + G : k0);              // unconditional, behind the conversions: see iir_lane_body
        asm volatile("" ::: "memory");
        double yd[G / 2];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const double x = xg[t];
            const double y = quad_perm_f64<0xA0>(__builtin_fma(b0, x, u[0]));      // quad_perm [0,0,2,2]: lane A's value
            const double z6 = quad_perm_f64<0xF5>(u[0]);                            // quad_perm [1,1,3,3]: lane B's first state, before its update
            const double tail = is_a ? z6 : 0.0;                                    // z_12 does not exist
#pragma unroll
            for (int sidx = 0; sidx + 1 < H; ++sidx) u[sidx] = __builtin_fma(ca[sidx], y, __builtin_fma(cb[sidx], x, u[sidx + 1]));
            u[H - 1] = __builtin_fma(ca[H - 1], y, __builtin_fma(cb[H - 1], x, tail));
            if (!(t & 1)) yd[t / 2] = y;                                            // decimate.py:41: samples 0, 2, 4, ...
        }
#pragma unroll
        for (int i = 0; i < G / 2; ++i) ydp[i] = yd[i];
    }
    if (xn && valid && is_a) {
#pragma unroll
        for (int i = 0; i < G / 4; ++i) *(double2*)(xn + ((L - G) >> 1) + 2 * i) = double2{ydp[2 * i], ydp[2 * i + 1]};
    }
    if (valid && q == a.nchunks - 1) {
#pragma unroll
        for (int sidx = 0; sidx < H; ++sidx) a.state[((size_t)c * a.nfilt + f) * kStates + H * half + sidx] = u[sidx];
    }
}

// grid (lane-split form): x = n_channels * ceil(nchunks / 21), y = band groups of 3 filters (a filter per lane, three lanes per chunk),
// then the decimator (a pair of lanes per chunk; its wavefronts are ceil(nchunks / 32) per channel: the others leave at once)
template <bool F32>
__global__ void __launch_bounds__(64) iir_lane_split_kernel(const IirStageArgs a, int n_band, int n_band_groups) {
    const int g = blockIdx.y, lane = threadIdx.x;
    if (g < n_band_groups) {
        const int f0 = g * 3, left = n_band - f0;
        iir_lane_body<1, 4, false, F32, 3>(a, f0, blockIdx.x, lane, left < 3 ? left : 3);
    } else {
        const int per_b = (a.nchunks + 20) / 21, per_d = (a.nchunks + 31) / 32;
        const int c = blockIdx.x / per_b, i = blockIdx.x - c * per_b;
        if (i >= per_d) return;
        iir_lane_dec_pair_body<F32>(a, c * per_d + i, lane);
    }
}

// grid: x = n_channels * ceil(nchunks / 64), y = filter groups (band filters kLaneBands at a time, then the decimator)
template <bool F32>
__device__ __forceinline__ void iir_lane_wave(const IirStageArgs& a, int n_band, int n_band_groups, int bx, int g, int lane) {
    if (g < n_band_groups) {
        const int f0 = g * kLaneBands, left = n_band - f0;
        if (left >= 3) iir_lane_body<3, 4, false, F32>(a, f0, bx, lane);
        else if (left == 2) iir_lane_body<2, 4, false, F32>(a, f0, bx, lane);
        else iir_lane_body<1, 4, false, F32>(a, f0, bx, lane);
    } else {
        iir_lane_body<1, 12, true, F32>(a, a.dec_filter, bx, lane);
    }
}

// The column form: a workgroup = `cols` chunk columns x every filter group of the launch (ngy wavefronts per column, at most
// kColWaves in all); the wavefronts of a column share its staged sample tiles (iir_lane_body, STAGED).  Wavefronts past the
// launch's last column shadow it and store nothing — every wavefront of a workgroup meets the same barriers.
constexpr int kColWaves = 9;
template <bool F32>
__global__ void __launch_bounds__(kColWaves * 64) iir_lane_col_kernel(const IirStageArgs a, int n_band, int n_band_groups, int nbx, int ngy, int cols) {
    extern __shared__ __attribute__((aligned(16))) char col_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wc = wave / ngy, g = wave - wc * ngy;
    const long long col = (long long)blockIdx.x * cols + wc;
    const bool live = col < nbx;
    const int bx = live ? (int)col : nbx - 1;
    char* lds = col_lds + (size_t)wc * 2 * ColTile<F32>::kBytes;
    const int tid_col = g * 64 + lane, nt_col = ngy * 64;
    if (g < n_band_groups) {
        const int f0 = g * kLaneBands, left = n_band - f0;
        if (left >= 3) iir_lane_body<3, 4, false, F32, 1, true>(a, f0, bx, lane, 3, lds, tid_col, nt_col, live);
        else if (left == 2) iir_lane_body<2, 4, false, F32, 1, true>(a, f0, bx, lane, 2, lds, tid_col, nt_col, live);
        else iir_lane_body<1, 4, false, F32, 1, true>(a, f0, bx, lane, 1, lds, tid_col, nt_col, live);
    } else {
        iir_lane_body<1, 12, true, F32, 1, true>(a, a.dec_filter, bx, lane, 1, lds, tid_col, nt_col, live);
    }
}

template <bool F32>
__global__ void __launch_bounds__(64) iir_lane_kernel(const IirStageArgs a, int n_band, int n_band_groups) {
    iir_lane_wave<F32>(a, n_band, n_band_groups, blockIdx.x, blockIdx.y, threadIdx.x);
}

// The same wavefronts as workgroups of FOUR, one per SIMD of a compute unit, with a dynamic LDS reservation (never touched) sized so
// that exactly ceil(workgroups / CUs) workgroups fit a CU: the hardware spreads a workgroup's waves over the four SIMDs and cannot
// stack more workgroups on one CU than on another.  A wavefront's filter group is its index modulo the number of groups, so the
// waves of a workgroup (and of a CU) are a mix of band groups and decimators in the launch's own proportion.
template <bool F32>
__global__ void __launch_bounds__(256) iir_lane_wg4_kernel(const IirStageArgs a, int n_band, int n_band_groups, int nbx, int ngy) {
    extern __shared__ char lane_reservation[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long v = (long long)blockIdx.x * 4 + wave;
    const int bx = (int)(v / ngy), g = (int)(v - (long long)bx * ngy);
    if (bx >= nbx) return;
    iir_lane_wave<F32>(a, n_band, n_band_groups, bx, g, threadIdx.x & 63);
}

// The lane kernel serves the output pass of an energy-only time-parallel stage when the stage is whole chunks of whole
// 4-sample groups on 16-byte boundaries, its filters are the bank's (4th-order bands in front of a 12th-order decimator)
// and nobody asked for band signals; everything else takes the slot kernel.
static bool lane_kernel_serves(const IirStageArgs& a, const int* orders) {
    if (!a.fused || a.pass != 2 || a.y != nullptr || a.nchunks < 1 || a.n != (long long)a.nchunks * a.chunk) return false;
    if (a.dec_filter != a.nfilt - 1 || orders[a.dec_filter] != 12) return false;
    for (int f = 0; f < a.dec_filter; ++f)
        if (orders[f] != 4 || a.band_index[f] < 0) return false;
    if (a.eblock && (a.eblock_len < 4 || a.chunk % a.eblock_len != 0)) return false;
    const bool vec_x = ((uintptr_t)a.x % 16 == 0) && (a.x_stride % (a.in_f32 ? 4 : 2) == 0);
    const bool vec_xn = !a.xnext || (((uintptr_t)a.xnext % 16 == 0) && (a.xnext_stride % 2 == 0));
    static const bool off = exp_env("FRT_IIR_NO_LANE_KERNEL") != nullptr;      // A/B runs
    return vec_x && vec_xn && !off;
}

enum LaneWhich { kWhichAll, kWhichBands, kWhichDec };      // the launch's filter groups: every one, the band groups, the decimator

// Does launch_iir_lane take the lane-split form (a filter per lane, the decimator on lane pairs) for this stage?  (Those bodies read
// their chunks' initial states from chunk_init: no look-back form.)
static bool lane_split_serves(const IirStageArgs& a, int n_channels, LaneWhich which) {
    const int n_band = a.dec_filter, groups = (n_band + kLaneBands - 1) / kLaneBands;
    const long long bx = (long long)n_channels * ((a.nchunks + 63) / 64);
    const int gy = which == kWhichAll ? groups + 1 : which == kWhichBands ? groups : 1;
    return which == kWhichAll && bx * gy <= (long long)exp_int("FRT_LANE_SPLIT_BELOW", device_cu_count());
}

static int launch_iir_lane(const IirStageArgs& a, int n_channels, hipStream_t stream, LaneWhich which) {
    const int n_band = a.dec_filter, groups = (n_band + kLaneBands - 1) / kLaneBands;
    const long long bx = (long long)n_channels * ((a.nchunks + 63) / 64);
    FRT_REQUIRE(bx < (1ll << 31), "iir lane pass: too many wavefronts");
    // the kernel takes its group from the launch's second axis: y < gb are band groups, the rest the decimator
    const int gy = which == kWhichAll ? groups + 1 : which == kWhichBands ? groups : 1;
    const int gb = which == kWhichDec ? 0 : groups;
    // The column form (round 6): the filter groups of a chunk column in one workgroup, samples staged through LDS.
    // Measured (profiles/r06_iir_lane_col.txt): it pays where the launch fills the chip and the column's wavefronts share enough samples
    // — 8 ch x 216 bands (nine wavefronts per column) 123.6 / 68.7 / 38.7 / 23.3 -> 114.2 / 61.8 / 35.3 / 20.9 us at stages 0-3, 8 ch x 27
    // bands (two per column) 120.0 / 71.1 / 37.9 -> 116.4 / 65.4 / 37.3 us at stages 0-2; with fewer workgroups than CUs a column's
    // wavefronts crowd one CU while others idle (216 bands, stages 4-6: 16.2 / 11.7 / 10.6 -> 18.3 / 16.8 / 17.1 us).  Tiles of 64
    // doubles instead of 32 (half the barriers, -DFRT_COL_F64_PIECES=32): equal at 216 bands, slower at 27 (two columns' 128 KB of LDS).
    const int cols = gy <= 2 ? 2 : 1;                            // at least four wavefronts per workgroup while a column has two
    const long long nwg = (bx + cols - 1) / cols;
    const int col_forced = option(kOptIirLaneColumns);
    const bool col_pays = nwg >= exp_int("FRT_LANE_COL_WGS", device_cu_count()) && (long long)gy * a.chunk >= exp_int("FRT_LANE_COL_SHARE", 512);
    if (which == kWhichAll && gy <= kColWaves && bx < (1ll << 30) && (col_forced > 0 || (col_forced < 0 && col_pays))) {
        static std::once_flag raised;                           // (handles may be driven from different threads)
        static hipError_t raise_rc = hipSuccess;
        std::call_once(raised, [] {
            raise_rc = hipFuncSetAttribute((const void*)iir_lane_col_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (raise_rc == hipSuccess)
                raise_rc = hipFuncSetAttribute((const void*)iir_lane_col_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        FRT_HIP_CHECK(raise_rc);
        size_t lds = (size_t)cols * 2 * (a.in_f32 ? ColTile<true>::kBytes : ColTile<false>::kBytes);
        // a launch of at most one workgroup per CU: a reservation of more than half of the CU's LDS keeps a second one off it (see below)
        if (nwg <= device_cu_count() && lds < (size_t)96 * 1024) lds = (size_t)96 * 1024;
        if (a.in_f32) hipLaunchKernelGGL(iir_lane_col_kernel<true>, dim3((unsigned)nwg), dim3(cols * gy * 64), lds, stream, a, n_band, gb, (int)bx, gy, cols);
        else hipLaunchKernelGGL(iir_lane_col_kernel<false>, dim3((unsigned)nwg), dim3(cols * gy * 64), lds, stream, a, n_band, gb, (int)bx, gy, cols);
        FRT_HIP_CHECK(hipGetLastError());
        return FRT_OK;
    }
    // Few wavefronts (at most one per CU in the form above: the low-rate stages, calls of a few channels): the lane-split form — a band
    // filter per lane on three lanes of a chunk, the decimator on a pair — has three / two times the wavefronts with a third / two
    // thirds of the dependent float64 instructions per sample each (iir_lane_split_kernel).  Measured (profiles/r05_iir_lane_split.txt),
    // it pays only while the chip is mostly idle: with 2.5 wavefronts per SIMD stage 0 of 8 ch x 27 bands takes 168 us against 136 (a
    // filter per LANE means coefficients in vector registers and permutes in the decimator's chain: more issue slots per sample
    // in total), but stages 6-8 go 9.8 / 10.0 / 10.6 -> 7.8 / 7.2 / 6.8 us and a 2-channel call 0.43 -> 0.365 ms.
    if (lane_split_serves(a, n_channels, which)) {
        const int sgroups = (n_band + 2) / 3;
        const long long sbx = (long long)n_channels * ((a.nchunks + 20) / 21);
        if (a.in_f32) hipLaunchKernelGGL(iir_lane_split_kernel<true>, dim3((unsigned)sbx, sgroups + 1), dim3(64), 0, stream, a, n_band, sgroups);
        else hipLaunchKernelGGL(iir_lane_split_kernel<false>, dim3((unsigned)sbx, sgroups + 1), dim3(64), 0, stream, a, n_band, sgroups);
        FRT_HIP_CHECK(hipGetLastError());
        return FRT_OK;
    }
    // A launch of at most one wavefront per SIMD with long chunks (the high-rate stages of a few channels: 8 channels x 27 bands at
    // chunks of 1024 are 512 band + 512 decimator wavefronts on 1024 SIMDs): as workgroups of four wavefronts, ONE per compute unit
    // (iir_lane_wg4_kernel) — single-wavefront workgroups end up two to a SIMD on part of the chip while other SIMDs idle, and a
    // SIMD shared by two of these float64 chains advances each at 0.73 of its solo rate (tools/exp/lane_probe.cpp).  Measured
    // (profiles/r05_iir_launches.txt): stage 0 160.6 -> 135.6 us, stage 1 72.2 -> 62.9 us; equal from chunks of 256 down, slower
    // with more than one workgroup per CU (216 bands).  (Only for launches that have the chip to themselves.)
    {
        const long long waves = bx * gy, nwg = (waves + 3) / 4;
        if (which == kWhichAll && nwg <= device_cu_count() && nwg * 2 > device_cu_count() && a.chunk >= 512 && !exp_env("FRT_LANE_NO_WG4")) {
            static std::once_flag raised;                   // (handles may be driven from different threads)
            static hipError_t raise_rc = hipSuccess;
            std::call_once(raised, [] {
                raise_rc = hipFuncSetAttribute((const void*)iir_lane_wg4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (raise_rc == hipSuccess)
                    raise_rc = hipFuncSetAttribute((const void*)iir_lane_wg4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            });
            FRT_HIP_CHECK(raise_rc);
            const size_t lds = (size_t)96 * 1024;            // more than half of a CU's 160 KB: a second workgroup does not fit
            if (a.in_f32) hipLaunchKernelGGL(iir_lane_wg4_kernel<true>, dim3((unsigned)nwg), dim3(256), lds, stream, a, n_band, gb, (int)bx, gy);
            else hipLaunchKernelGGL(iir_lane_wg4_kernel<false>, dim3((unsigned)nwg), dim3(256), lds, stream, a, n_band, gb, (int)bx, gy);
            FRT_HIP_CHECK(hipGetLastError());
            return FRT_OK;
        }
    }
    if (a.in_f32) hipLaunchKernelGGL(iir_lane_kernel<true>, dim3((unsigned)bx, gy), dim3(64), 0, stream, a, n_band, gb);
    else hipLaunchKernelGGL(iir_lane_kernel<false>, dim3((unsigned)bx, gy), dim3(64), 0, stream, a, n_band, gb);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}

// ---- time-parallel mode, pass 1: zero-state end states as dot products -----------------------------------
// The end state of a chunk started from zero is linear in its samples:  s = sum_k A^(L-1-k) B x[k],
// B[s] = b[s+1] - a[s+1] b[0].  Unlike the recurrence this has no serial dependency: a lane owns one
// (channel, chunk) column and walks its samples, the table row g[k][0..rows) is wave-uniform (scalar loads,
// the FMAs take it from SGPRs), 32 states per lane are accumulated at a time.  blockIdx.y splits the chunk
// into K-slices whose partial sums the scan kernel adds; blockIdx.z tiles the rows.
constexpr int kZsRows = 32;             // the table's rows are padded to a multiple of this
#ifndef FRT_ZS_ROWS
#define FRT_ZS_ROWS 16
#endif
#ifndef FRT_ZS_COLS
#define FRT_ZS_COLS 2
#endif
constexpr int kZsRowsPerPass = FRT_ZS_ROWS, kZsCols = FRT_ZS_COLS;      // states x columns per lane (see iir_zero_state_kernel)
static_assert(kZsRows % kZsRowsPerPass == 0, "row groups tile the padded table");
#ifndef FRT_ZS_ABLATE           // experiment builds (wrong results): 1 one table row for every sample, 2 no sample loads
#define FRT_ZS_ABLATE 0
#endif
#ifndef FRT_ZS_BATCH            // samples per lane and batch of the zero-state kernel (8, 16 or 32)
#define FRT_ZS_BATCH 8
#endif
constexpr int kMaxSlices = 8;
#ifndef FRT_IIR_LOOKBACK_MAX
#define FRT_IIR_LOOKBACK_MAX 16
#endif
constexpr int kLookbackMax = FRT_IIR_LOOKBACK_MAX;      // chunks an output pass looks back over instead of waiting for a scan launch

struct ZeroStateArgs {
    const void* x;
    long long x_stride;
    int n, in_f32;
    int chunk, nchunks, n_channels, nfilt;
    int slice;                 // samples per K-slice (multiple of 8)
    int vec;                   // rows 16-byte aligned: vector loads allowed
    const double* table;       // [chunk][rows_padded]: g[k][r]
    const double* table_m;     // the same values in the MFMA kernel's operand order (zs_mfma_index)
    int rows, rows_padded;
    const int* rowmap;         // [rows_padded]: f * kStates + s of each row, -1 for padding
    const int* order;          // [nfilt]: a filter's order decides how its chunk states are packed (chunk_state_stride)
    double* partial;           // [n_slices][C][nfilt][nchunks][kStates]
    long long partial_stride;
    int rt_base, rt_count;     // MFMA kernel: the launch serves row tiles rt_base .. rt_base + rt_count - 1 (tile 0: the decimator)
    const double* snap_src;    // MFMA kernel: the stage's carried state is copied to snap_dst (snap_count doubles) for the look-back
    double* snap_dst;          // form of the output pass (IirStageArgs::state_in); null: no copy
    int snap_count;
};

// ROWS states x COLS (channel, chunk) columns per lane.  The table row g[k][.] reaches the FMAs through scalar registers,
// 8 bytes of scalar-cache traffic per FMA instruction when a lane owns one column — and that stream, not the arithmetic,
// was what bound the kernel (a build reading one table row for every k ran 3x faster).  Every g[k][r] now feeds COLS FMAs.
template <int ROWS, int COLS>
__global__ void __launch_bounds__(64) iir_zero_state_kernel(const ZeroStateArgs a) {
    const int lane = threadIdx.x;
    const long long ncols = (long long)a.n_channels * a.nchunks;
    const int r0 = blockIdx.z * ROWS;
    const int k0 = blockIdx.y * a.slice;
    bool valid[COLS];
    int ch[COLS], cq[COLS];
    long long first[COLS], xrow[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const long long col = ((long long)blockIdx.x * 64 + lane) * COLS + j;
        valid[j] = col < ncols;
        const long long cc = valid[j] ? col : 0;
        ch[j] = (int)(cc / a.nchunks);
        cq[j] = (int)(cc % a.nchunks);
        first[j] = (long long)cq[j] * a.chunk + k0;                    // stage sample index of the column's first sample
        xrow[j] = (long long)ch[j] * a.x_stride;
    }
    const double* __restrict__ g = a.table + (size_t)k0 * a.rows_padded + r0;
    double acc[COLS][ROWS];
#pragma unroll
    for (int j = 0; j < COLS; ++j)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[j][r] = 0.0;
    // KB samples of each of the lane's streams per batch, their loads in flight together; 16-byte loads when the rows are
    // aligned (a.vec) — a quarter / half of the load instructions, each of which touches 64 different cache lines
    constexpr int KB = FRT_ZS_BATCH;
    for (int k = 0; k < a.slice; k += KB) {
        double xv[COLS][KB];
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            const long long i0 = first[j] + k;
            if (FRT_ZS_ABLATE & 2) {
#pragma unroll
                for (int u = 0; u < KB; ++u) xv[j][u] = (double)(k + u);
            } else if (a.vec && valid[j] && i0 + KB <= a.n && k + KB <= a.slice) {
                if (a.in_f32) {
                    const float4* p = (const float4*)((const float*)a.x + xrow[j] + i0);
                    float4 raw[KB / 4];
#pragma unroll
                    for (int u = 0; u < KB / 4; ++u) raw[u] = p[u];
#pragma unroll
                    for (int u = 0; u < KB / 4; ++u) {
                        xv[j][4 * u] = raw[u].x; xv[j][4 * u + 1] = raw[u].y; xv[j][4 * u + 2] = raw[u].z; xv[j][4 * u + 3] = raw[u].w;
                    }
                } else {
                    const double2* p = (const double2*)((const double*)a.x + xrow[j] + i0);
#pragma unroll
                    for (int u = 0; u < KB / 2; ++u) {
                        const double2 v = p[u];
                        xv[j][2 * u] = v.x;
                        xv[j][2 * u + 1] = v.y;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const long long i = i0 + u;
                    const bool ok = valid[j] && i < a.n && k + u < a.slice;
                    const long long ii = ok ? xrow[j] + i : 0;
                    const double v = a.in_f32 ? (double)((const float*)a.x)[ii] : ((const double*)a.x)[ii];
                    xv[j][u] = ok ? v : 0.0;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (k + u < a.slice) {                                   // uniform; a.slice is a multiple of 8
                const double* __restrict__ gk = g + (size_t)((FRT_ZS_ABLATE & 1) ? 0 : k + u) * a.rows_padded;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const double gv = gk[r];
#pragma unroll
                    for (int j = 0; j < COLS; ++j) acc[j][r] = __builtin_fma(gv, xv[j][u], acc[j][r]);
                }
            }
        }
    }
    double* out = a.partial + (size_t)blockIdx.y * a.partial_stride;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        if (!valid[j]) continue;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int m = a.rowmap[r0 + r];                               // uniform
            if (m >= 0) out[((size_t)ch[j] * a.nfilt + (m >> 4)) * a.nchunks * kStates + (size_t)cq[j] * chunk_state_stride(a.order[m >> 4]) + (m & 15)] = acc[j][r];
        }
    }
}

// The same product on the matrix cores.  It IS a dense contraction — end[row][col] = sum_k g[k][row] x[col][k], 24-108 rows,
// 64-16384 samples deep, C x chunks columns — and on the vector ALUs it was bound by operand delivery, not arithmetic: one
// table value per FMA through the scalar cache (above).  v_mfma_f64_16x16x4_f64 takes A = g^T (16 rows x 4 samples) and
// B = x (4 samples x 16 columns) from vector registers, one value per lane each, and reuses them 16-fold inside the array.
//   lane l = (j = l % 16, g = l / 16).  A block of 16 samples is four MFMA steps t = 0..3; in step t slot g stands for sample
//   16 kb + 4 g + t (the sum over k does not care which slot a sample sits in, as long as A and B agree).  So a lane needs
//   four CONSECUTIVE samples of column j per block — one 16-byte load, the four lanes of a column covering 64 contiguous
//   bytes — and, per row tile, the four table values g[16 kb + 4 g + t][16 rt + j], stored contiguously for exactly this
//   (zs_mfma_index): 32 bytes per lane, 2 KB contiguous per wavefront.
//   D: lane (j, g) holds rows 16 rt + 4 v + g (v = 0..3) of column j (tools/exp/mfma_f64_layout.cpp prints the layout).
typedef double zs_double4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline size_t zs_mfma_index(int k, int row, int row_tiles) {
    return (((((size_t)(k >> 4) * row_tiles + (row >> 4)) * 4 + ((k >> 2) & 3)) * 16 + (row & 15)) * 4) + (k & 3);
}
#ifndef FRT_ZS_UNROLL
#define FRT_ZS_UNROLL 2
#endif

// kZsTiles row tiles (of 16 rows) per workgroup: 2 (every x value then feeds two MFMAs), or 1 for a range of a single tile
template <int kZsTiles>
__global__ void __launch_bounds__(64) iir_zero_state_mfma_kernel(const ZeroStateArgs a) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    if (a.snap_dst && blockIdx.y == 0 && blockIdx.z == 0)                 // (the launch's first plane of workgroups shares the copy)
        for (long long i = (long long)blockIdx.x * 64 + lane; i < a.snap_count; i += (long long)gridDim.x * 64) a.snap_dst[i] = a.snap_src[i];
    const long long ncols = (long long)a.n_channels * a.nchunks;
    const long long col = (long long)blockIdx.x * 16 + j;
    const bool valid = col < ncols;
    const long long cc = valid ? col : 0;
    const int c = (int)(cc / a.nchunks), q = (int)(cc % a.nchunks);
    const int k0 = blockIdx.y * a.slice;                                  // a multiple of 16
    const int rt0 = a.rt_base + blockIdx.z * kZsTiles, row_tiles = a.rows_padded / 16;
    const long long first = (long long)q * a.chunk + k0 + 4 * g;          // the lane's first sample
    const long long xrow = (long long)c * a.x_stride;
    zs_double4 acc[kZsTiles];
#pragma unroll
    for (int rt = 0; rt < kZsTiles; ++rt) acc[rt] = zs_double4{0.0, 0.0, 0.0, 0.0};
    const zs_double4* __restrict__ tm = (const zs_double4*)a.table_m;
    constexpr int UN = FRT_ZS_UNROLL;                                                 // blocks of 16 samples whose loads are in flight together
    // Every column of the wavefront a whole slice of an aligned row (the bulk of every launch): a loop without the per-block choices
    // below.  With them the float32 samples of stage 0 were CONVERTED inside the branch that had loaded them — a wait for every load
    // right behind it, the table loads and the next block's samples queued behind that wait (ISA: global_load_dwordx4, s_waitcnt vmcnt(0),
    // v_cvt_f64_f32 x 4, then the table loads): the blocks' loads were never in flight together at the stage that has the most of them.
    // Same products in the same order: bit-identical end states.
    const bool whole = a.vec && valid && first - 4 * g + a.slice <= a.n && (a.slice / 16) % UN == 0;
    if (__all(whole)) {
        const int tile0 = rt0 < row_tiles ? rt0 : row_tiles - 1, tile1 = rt0 + 1 < row_tiles ? rt0 + 1 : row_tiles - 1;
        const size_t kb0 = (size_t)(k0 >> 4);
        auto table_at = [&](size_t kblock, int rt) {
            if (FRT_ZS_ABLATE & 1) kblock = kb0;                    // (timing experiments: every block the same table values — from L1)
            return tm[((kblock * row_tiles + (rt == 0 ? tile0 : tile1)) * 4 + g) * 16 + j];
        };
        if (a.in_f32) {
            const float* xp = (const float*)a.x + xrow + first;
            for (int kb = 0; kb < a.slice / 16; kb += UN) {
                float4 raw[UN];
                zs_double4 av[UN][kZsTiles];
#pragma unroll
                for (int u = 0; u < UN; ++u) raw[u] = (FRT_ZS_ABLATE & 2) ? float4{0.5f, 0.25f, -0.5f, 1.f} : *(const float4*)(xp + 16 * (kb + u));
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int rt = 0; rt < kZsTiles; ++rt) av[u][rt] = table_at(kb0 + kb + u, rt);
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const double xs[4] = {(double)raw[u].x, (double)raw[u].y, (double)raw[u].z, (double)raw[u].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int rt = 0; rt < kZsTiles; ++rt)
                            acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][rt][t], xs[t], acc[rt], 0, 0, 0);
                }
            }
        } else {
            const double* xp = (const double*)a.x + xrow + first;
            for (int kb = 0; kb < a.slice / 16; kb += UN) {
                double2 raw[UN][2];
                zs_double4 av[UN][kZsTiles];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    raw[u][0] = *(const double2*)(xp + 16 * (kb + u));
                    raw[u][1] = *(const double2*)(xp + 16 * (kb + u) + 2);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int rt = 0; rt < kZsTiles; ++rt) av[u][rt] = table_at(kb0 + kb + u, rt);
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const double xs[4] = {raw[u][0].x, raw[u][0].y, raw[u][1].x, raw[u][1].y};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int rt = 0; rt < kZsTiles; ++rt)
                            acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][rt][t], xs[t], acc[rt], 0, 0, 0);
                }
            }
        }
    } else
    for (int kb = 0; kb < a.slice / 16; kb += UN) {
        double xs[UN][4];
        zs_double4 av[UN][kZsTiles];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool in_slice = kb + u < a.slice / 16;
            const long long i0 = first + 16 * (kb + u);
            if (in_slice && a.vec && valid && i0 + 4 <= a.n) {
                if (a.in_f32) {
                    const float4 v = *(const float4*)((const float*)a.x + xrow + i0);
                    xs[u][0] = v.x; xs[u][1] = v.y; xs[u][2] = v.z; xs[u][3] = v.w;
                } else {
                    const double2* p = (const double2*)((const double*)a.x + xrow + i0);
                    const double2 lo = p[0], hi = p[1];
                    xs[u][0] = lo.x; xs[u][1] = lo.y; xs[u][2] = hi.x; xs[u][3] = hi.y;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long long i = i0 + t;
                    const bool ok = in_slice && valid && i < a.n;
                    const long long ii = ok ? xrow + i : 0;
                    const double v = a.in_f32 ? (double)((const float*)a.x)[ii] : ((const double*)a.x)[ii];
                    xs[u][t] = ok ? v : 0.0;
                }
            }
            const size_t kblock = (size_t)(k0 >> 4) + (in_slice ? kb + u : 0);
#pragma unroll
            for (int rt = 0; rt < kZsTiles; ++rt) {
                const int tile = rt0 + rt < row_tiles ? rt0 + rt : row_tiles - 1;      // (an odd range's last workgroup: its second tile is not stored)
                av[u][rt] = tm[((kblock * row_tiles + tile) * 4 + g) * 16 + j];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < kZsTiles; ++rt)
                    acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][rt][t], xs[u][t], acc[rt], 0, 0, 0);
    }
    if (!valid) return;
    double* out = a.partial + (size_t)blockIdx.y * a.partial_stride;
#pragma unroll
    for (int rt = 0; rt < kZsTiles; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (rt0 + rt >= a.rt_base + a.rt_count) continue;              // the second tile of an odd range's last workgroup
            const int m = a.rowmap[(rt0 + rt) * 16 + 4 * v + g];
            if (m >= 0) out[((size_t)c * a.nfilt + (m >> 4)) * a.nchunks * kStates + (size_t)q * chunk_state_stride(a.order[m >> 4]) + (m & 15)] = acc[rt][v];
        }
}

// partial[0][i] += partial[1..n_slices)[i]: the K-slices of pass 1 summed in slice order
__global__ void __launch_bounds__(256) iir_slice_sum_kernel(double* __restrict__ partial, long long stride, int n_slices, long long count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double e = partial[i];
    for (int p = 1; p < n_slices; ++p) e += partial[(size_t)p * stride + i];
    partial[i] = e;
}

// the same for the filters f0 .. f0 + nf - 1 only (the layout is [C][nfilt][nchunks][kStates]): grid (x over nf x nchunks x kStates, C)
__global__ void __launch_bounds__(256) iir_slice_sum_range_kernel(double* __restrict__ partial, long long stride, int n_slices, int nfilt,
                                                                  int nchunks, int f0, int nf) {
    const long long per = (long long)nchunks * kStates, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per * nf) return;
    const long long at = ((long long)blockIdx.y * nfilt + f0) * per + i;
    double e = partial[at];
    for (int p = 1; p < n_slices; ++p) e += partial[(size_t)p * stride + at];
    partial[at] = e;
}

// A^L z for the states of a filter held by the lanes of a DPP row, lane s owning row s of the matrix.  NT: number of
// (leading) non-zero columns, i.e. the filter order rounded up to 4; four interleaved partial sums (the serial part
// of the scan is this dependency chain).
#ifndef FRT_SCAN_ABLATE          // experiment builds: 1 = no arithmetic in the walks (z <- e + z), 2 = no end-state loads (timing only: wrong results)
#define FRT_SCAN_ABLATE 0
#endif
// lane T of every quad, to its quad
template <int T>
__device__ __forceinline__ double dpp_quad_bcast(double v) {
    constexpr int ctrl = T | (T << 2) | (T << 4) | (T << 6);       // quad_perm [T, T, T, T]
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// NT: the filter's order rounded up to 4; a 4th-order filter's row is a QUAD of lanes (round 6: its 16-lane row had twelve idle lanes —
// 24 of a 1/24-octave stage's 25 filters), everything above it a 16-lane DPP row
template <int NT>
__device__ __forceinline__ double row_matvec(const double (&m)[NT], double z) {
    if (FRT_SCAN_ABLATE & 1) return z * m[0];
    double p[4] = {0.0, 0.0, 0.0, 0.0};
    static_assert(kStates == 16 && NT % 4 == 0 && NT <= 16, "one 16-lane row or one quad per filter");
    if constexpr (NT == 4)
        return __builtin_fma(m[1], dpp_quad_bcast<1>(z), m[0] * dpp_quad_bcast<0>(z)) + __builtin_fma(m[3], dpp_quad_bcast<3>(z), m[2] * dpp_quad_bcast<2>(z));
    // (contracted: the scan belongs to the time-parallel mode, which re-associates the recurrence anyway)
#define FRT_SCAN_TERM(T) if (T < NT) p[(T) & 3] = __builtin_fma(m[T], dpp_row_bcast<T>(z), p[(T) & 3]);
    FRT_SCAN_TERM(0) FRT_SCAN_TERM(1) FRT_SCAN_TERM(2) FRT_SCAN_TERM(3) FRT_SCAN_TERM(4) FRT_SCAN_TERM(5)
    FRT_SCAN_TERM(6) FRT_SCAN_TERM(7) FRT_SCAN_TERM(8) FRT_SCAN_TERM(9) FRT_SCAN_TERM(10) FRT_SCAN_TERM(11)
    FRT_SCAN_TERM(12) FRT_SCAN_TERM(13) FRT_SCAN_TERM(14) FRT_SCAN_TERM(15)
#undef FRT_SCAN_TERM
    return (p[0] + p[1]) + (p[2] + p[3]);
}

// z_{q+1} = A^L z_q + s_q over the chunks of one (channel, filter): chunk_end (s_q, the zero-state end states of pass 1)
// -> what the output pass needs to give every chunk its true initial state.
//
// The chunks are cut into scan ROWS of `group` consecutive chunks; a row is a 16-lane DPP row (lane s = state s) that runs
// its chunks from a ZERO state and records, per chunk, the state it would start from had its row started at zero
// (chunk_init) and its own end state E[r].  The true state at the start of row r is S[r] = E[r-1] + M S[r-1], M = A^(L group).
// Rounds 1-2 chained that recurrence serially (row 0 of a single workgroup walked all 32 rows of a (channel, filter), the
// rows being 128 chunks long: 160 serial steps of ~0.3 us, 52 us per stage with 32 of 256 CUs busy — after the lane-per-
// chunk output pass the largest item of the bank).  But the filters are stable: the host picks `group` so that M^2 has
// vanished in float64 (max |M^2| < 1e-20: L group >= ~3000 samples for the decimator's pole radius 0.9923), and then
//     S[r] = E[r-1] + M E[r-2]          (r >= 2;  S[1] = E[0] + M S[0];  S[0] = the carried state)
// is the same sum to the last bit that matters — no chain: every row depends on its two predecessors only, rows are a few
// chunks long at the high-rate stages (4 chunks of 1024 samples) and all of them run at once.  A workgroup owns 30
// consecutive rows and re-runs the two rows in front of them (halo) so that it needs nobody else's end states.
// A chunk's initial state is chunk_init[q] + (A^L)^(q - row start) S[row]: the owned rows form it at the end of this kernel
// (until round 4 the output pass did, lane by lane).  power_l / power_g: [nfilt][16][16] row-major A^L and M.
// Round 4: a row is at most kScanRowMax (16) chunks.  At the low-rate stages a chunk is 64 samples and the decay takes 64 chunks: rows
// of 64 were 64 dependent steps of ~0.3 us, 20 us per stage for a few KB of states — the largest launch of stages 4-8.  Rows of
// 16 chunks need more than two predecessors: S[r] = sum_{k=1..K} Mr^(k-1) E[r-k], K = the rows the decay spans (Horner: K - 1
// dependent products, all rows at once), with E[-1] = the carried state.  16 + 7 dependent steps instead of 64 + 1.
constexpr int kScanRows = 32;             // rows per workgroup: `halo` of them re-run their predecessors', the rest are owned
#ifndef FRT_SCAN_ROW_MAX
#define FRT_SCAN_ROW_MAX 16
#endif
constexpr int kScanRowMax = FRT_SCAN_ROW_MAX;      // measured 4 / 8 / 16: see tools/exp/README.md
#ifndef FRT_SCAN_BATCH
#define FRT_SCAN_BATCH 8
#endif
constexpr int kScanBatch = FRT_SCAN_BATCH;         // end states requested per trip of a row's walk

template <int NT>
__device__ __forceinline__ void iir_scan_body(const double* __restrict__ power_l, const double* __restrict__ power_g,
                                              const double* __restrict__ state, const double* __restrict__ chunk_end,
                                              double* __restrict__ chunk_init, int gid, int seg, int f,
                                              bool live, int nchunks, int group, int nrows, int halo, double (*gend)[kStates]) {
    constexpr int LPR = NT == 4 ? 4 : 16;                      // lanes per row
    const int row = threadIdx.x / LPR, s = threadIdx.x & (LPR - 1);
    const int r = seg * (kScanRows - halo) - halo + row;      // global row; the first `halo` rows of the workgroup are the halo
    const bool row_ok = r >= 0 && r < nrows;
    const bool owned = row >= halo && row_ok;
    double m[NT], mg[NT];                                      // (both requested up front: the second table is needed after the rows' walk)
#pragma unroll
    for (int t = 0; t < NT; ++t) m[t] = power_l[((size_t)f * kStates + s) * kStates + t];
#pragma unroll
    for (int t = 0; t < NT; ++t) mg[t] = power_g[((size_t)f * kStates + s) * kStates + t];
    const double s0 = live ? state[(size_t)gid * kStates + s] : 0.0;
    constexpr int CS = chunk_state_stride(NT);                // doubles per chunk of this filter's block
    const double* ce = chunk_end + (size_t)gid * nchunks * kStates + s;
    double* ci = chunk_init + (size_t)gid * nchunks * kStates + s;
    const int q0 = row_ok ? r * group : 0;
    const int q1 = row_ok ? ((q0 + group) < nchunks ? (q0 + group) : nchunks) : 0;
    // zero-state end states of chunks q .. q+7; lanes above the order stay 0 whatever the scratch holds.  (Summing the
    // K-slices of the table product here instead of in a launch of their own was measured in round 3: slower — the rows'
    // strided reads of four slices, the halo rows' included, cost more than the streaming kernel's 5 us.)
    auto end_states = [&](int q, double (&e)[kScanBatch]) {
#pragma unroll
        for (int j = 0; j < kScanBatch; ++j) e[j] = (live && q + j < q1 && !(FRT_SCAN_ABLATE & 2)) ? ce[(size_t)(q + j) * CS] : 0.0;
    };
    // the row's chunks from a zero state.  Rows of at most kScanRowMax chunks (every shape but a workgroup short of halo rows) keep
    // their end states in registers for the second walk below.
    constexpr int kKeep = kScanRowMax / kScanBatch;
    static_assert(kScanRowMax % kScanBatch == 0, "a kept row is whole batches");
    const bool keep = group <= kScanRowMax;                     // uniform
    double ek[kKeep][kScanBatch];
    double z = 0.0;
    if (keep) {
#pragma unroll
        for (int b = 0; b < kKeep; ++b) end_states(q0 + b * kScanBatch, ek[b]);
#pragma unroll
        for (int b = 0; b < kKeep; ++b) {
#pragma unroll
            for (int j = 0; j < kScanBatch; ++j)
                if (q0 + b * kScanBatch + j < q1) z = ek[b][j] + row_matvec<NT>(m, z);
        }
    } else {
        for (int q = q0; q < q1; q += kScanBatch) {
            double e[kScanBatch];
            end_states(q, e);
#pragma unroll
            for (int j = 0; j < kScanBatch; ++j) {
                if (q + j < q1) z = e[j] + row_matvec<NT>(m, z);
            }
        }
    }
    gend[row][s] = z;
    __syncthreads();
    // the true state at the start of every owned row, from its two predecessors (or the carried state next to the start)
    // E[r - k]: the workgroup's own rows (k <= row), the carried state for the virtual row -1, zero before it
    auto end_of = [&](int k) -> double { return r - k >= 0 ? (k <= row ? gend[row - k][s] : 0.0) : (r - k == -1 ? s0 : 0.0); };
    double start = end_of(halo);
    for (int k = halo - 1; k >= 1; --k) start = end_of(k) + row_matvec<NT>(mg, start);
    // The true initial state of every chunk of an owned row: the row is walked once more, from its true start — the same
    // z <- s_q + A^L z, 16 dependent steps at most, the table already in registers.  (Until round 4 the output pass composed it lane
    // by lane as prefix + (A^L)^i S[r] from a table of the in-row powers: per-lane gathers of a 16 x 16 matrix, 13-15 dependent
    // round trips, most of what a low-rate stage's launch lasted.  Rows were 128 chunks long when the replay was dropped in round 2.)
    if (owned) {
        double zt = start;
        if (keep) {
#pragma unroll
            for (int b = 0; b < kKeep; ++b) {
#pragma unroll
                for (int j = 0; j < kScanBatch; ++j) {
                    const int q = q0 + b * kScanBatch + j;
                    if (q < q1) {
                        ci[(size_t)q * CS] = zt;
                        zt = ek[b][j] + row_matvec<NT>(m, zt);
                    }
                }
            }
        } else {
            for (int q = q0; q < q1; q += kScanBatch) {
                double e[kScanBatch];
                end_states(q, e);
#pragma unroll
                for (int j = 0; j < kScanBatch; ++j) {
                    if (q + j < q1) {
                        ci[(size_t)(q + j) * CS] = zt;
                        zt = e[j] + row_matvec<NT>(m, zt);
                    }
                }
            }
        }
    }
}

// grid.x = (channel, filter) pairs x segments of kScanOwned rows
__global__ void __launch_bounds__(kScanRows * 16) iir_scan_kernel(const double* __restrict__ power_l,
                                                                  const double* __restrict__ power_g,
                                                                  const double* __restrict__ state,
                                                                  const double* __restrict__ chunk_end,
                                                                  const int* __restrict__ order,
                                                                  double* __restrict__ chunk_init, int nfilt,
                                                                  int nchunks, int group, int nrows, int nseg, int halo, int f0, int nf) {
    __shared__ double gend[kScanRows][kStates];
    // the launch serves filters f0 .. f0 + nf - 1 of every channel; gid: the (channel, filter) pair's index in the [C][nfilt] layout
    const int lid = blockIdx.x / nseg, seg = blockIdx.x - lid * nseg;
    const int f = f0 + lid % nf, gid = (lid / nf) * nfilt + f;
    const int ord = order[f];                                 // uniform in the workgroup
    if (ord <= 4 && threadIdx.x >= kScanRows * 4) return;     // a 4th-order filter's rows are quads: the first two wavefronts hold all 32
                                                              // (whole wavefronts leave: the barriers below count the ones that stay)
    const bool live = ord <= 4 ? (int)(threadIdx.x & 3) < ord : (int)(threadIdx.x & 15) < ord;
    if (ord <= 4) iir_scan_body<4>(power_l, power_g, state, chunk_end, chunk_init, gid, seg, f, live, nchunks, group, nrows, halo, gend);
    else if (ord <= 12) iir_scan_body<12>(power_l, power_g, state, chunk_end, chunk_init, gid, seg, f, live, nchunks, group, nrows, halo, gend);
    else iir_scan_body<16>(power_l, power_g, state, chunk_end, chunk_init, gid, seg, f, live, nchunks, group, nrows, halo, gend);
}

// (Round 6: the same scan with sixteen rows per wavefront as the columns of Z <- E + P Z on v_mfma_f64_16x16x4_f64 was built, parity-
// green against this kernel, and measured SLOWER — 15-22 us per launch against 10-19 (profiles/r06_iir_mfma_scan.txt).  A float64 MFMA
// occupies gfx950's matrix pipe for 64 cycles, four dependent ones per step cost what the twelve broadcast + multiply-add pairs of a
// DPP row cost, and what a scan launch lasts is its fixed part — launch, the tables' and end states' round trips, the exchange through
// LDS, the stores: ~10 us with either arithmetic.)

// sp_blk = E_blk + sp_{blk-1} * (1-alpha)^n  (exp_smoothing.py:52-54), optional dB + weighting.
// One workgroup per channel.  The recurrence is two dependent operations per block; what costs is fetching
// the block energies, so they go through LDS in tiles: all threads load a tile (coalesced, the next tile's
// loads already in flight), `nbands` threads run the recurrence over it in LDS, all threads convert and write.
constexpr int kEnergyThreads = 256;
constexpr int kEnergyPerThread = 8;                                   // tile = 2048 values
static_assert((kMaxFilters - 1) * kNOctave <= kEnergyThreads, "one recurrence thread per band");

// `sub`: the block axis is `sub` times finer than the caller's blocks (a time-parallel chunk shorter than the block, see
// frt_octbank_energies): the recurrence runs over every entry, every sub-th value is an output.
__global__ void __launch_bounds__(kEnergyThreads) energy_scan_kernel(const double* __restrict__ eblock,
                                                                     const double* __restrict__ decay_n, double* __restrict__ smooth,
                                                                     void* __restrict__ out, int out_f32, int nblocks, int nbands,
                                                                     const double* __restrict__ weight_db, int as_db, int sub) {
    __shared__ double tile[kEnergyThreads * kEnergyPerThread];
    __shared__ double seg_end[kEnergyThreads], seg_pow[kEnergyThreads], seg_carry[kEnergyThreads];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int tile_blocks = kEnergyThreads * kEnergyPerThread / nbands;      // whole blocks per tile (nbands <= 256)
    const int tile_vals = tile_blocks * nbands;
    const double* src = eblock + (size_t)c * nblocks * nbands;
    const size_t obase = (size_t)c * (nblocks / sub) * nbands;
    double prev = tid < nbands ? smooth[(size_t)c * nbands + tid] : 0.0;
    double pre[kEnergyPerThread];
    auto fetch = [&](int b0) {
        const long long left = (long long)(nblocks - b0) * nbands;
#pragma unroll
        for (int j = 0; j < kEnergyPerThread; ++j) {
            const int i = tid + j * kEnergyThreads;
            pre[j] = (i < tile_vals && i < left) ? src[(size_t)b0 * nbands + i] : 0.0;
        }
    };
    fetch(0);
    for (int b0 = 0; b0 < nblocks; b0 += tile_blocks) {
        const int nb = (nblocks - b0) < tile_blocks ? (nblocks - b0) : tile_blocks;
#pragma unroll
        for (int j = 0; j < kEnergyPerThread; ++j) tile[tid + j * kEnergyThreads] = pre[j];
        __syncthreads();
        if (b0 + tile_blocks < nblocks) fetch(b0 + tile_blocks);
        // The recurrence over the tile's blocks, split into `nseg` runs of consecutive blocks per band so that
        // nseg * nbands threads work: each run from a zero carry, the runs chained, then the carry folded in
        // (sp_b = local_b + carry d^(b - run start + 1)).
        const int nseg = kEnergyThreads / nbands > 8 ? 8 : (kEnergyThreads / nbands < 1 ? 1 : kEnergyThreads / nbands);
        const int seg_len = (nb + nseg - 1) / nseg;
        const int seg = tid / nbands, band = tid - seg * nbands;
        const bool worker = seg < nseg;
        const double dk = worker ? decay_n[band] : 0.0;
        const int sb0 = seg * seg_len, sb1 = (sb0 + seg_len) < nb ? (sb0 + seg_len) : nb;
        double local = 0.0, dpow = 1.0;
        if (worker) {
            for (int b = sb0; b < sb1; ++b) {
                local = tile[b * nbands + band] + local * dk;
                tile[b * nbands + band] = local;
                dpow *= dk;
            }
            seg_end[seg * nbands + band] = local;
            seg_pow[seg * nbands + band] = dpow;
        }
        __syncthreads();
        if (tid < nbands) {
            double carry = prev;
            for (int g = 0; g < nseg; ++g) {
                seg_carry[g * nbands + tid] = carry;
                carry = seg_end[g * nbands + tid] + carry * seg_pow[g * nbands + tid];
            }
            prev = carry;
        }
        __syncthreads();
        if (worker) {
            const double carry = seg_carry[seg * nbands + band];
            double p = dk;
            for (int b = sb0; b < sb1; ++b) {
                tile[b * nbands + band] += carry * p;
                p *= dk;
            }
        }
        __syncthreads();
        const int vals = nb * nbands;
        for (int i = tid; i < vals; i += kEnergyThreads) {
            const int bi = b0 + i / nbands, band = i % nbands;
            if ((bi + 1) % sub != 0) continue;
            double v = tile[i];
            if (as_db) v = 10.0 * log10(v + 1e-30) + (weight_db ? weight_db[band] : 0.0);
            const size_t o = obase + (size_t)(bi / sub) * nbands + band;
            if (out_f32) ((float*)out)[o] = (float)v;
            else ((double*)out)[o] = v;
        }
        __syncthreads();
    }
    if (tid < nbands) smooth[(size_t)c * nbands + tid] = prev;
}

// The same recurrence for long batches, split along time: the single-workgroup-per-channel kernel above walks a batch of
// 4096 blocks in 55 tiles of four barriers each — 143 us of an 1.8 ms bank, with 8 of 256 CUs busy.  Here a workgroup owns
// kEnergySplit consecutive blocks of one channel, a thread one band: (1) energy_local_kernel runs the recurrence from a
// zero carry inside every split (in place) and records the split's last value; (2) energy_finish_kernel chains the splits
// before its own (sp = end_g + carry d^len, a few dozen steps), folds the carry in (local_b + carry d^(b - b0 + 1)),
// converts and writes.
constexpr int kEnergySplit = 64;
#ifndef FRT_ENERGY_BATCH
#define FRT_ENERGY_BATCH 8
#endif
#ifndef FRT_ENERGY_CHAIN
#define FRT_ENERGY_CHAIN 8
#endif
// Values a thread requests per trip.  Both kernels are one or four wavefronts per workgroup walking strided columns: what they
// cost is their dependent memory round trips, so the next trip's values are requested before this trip's are used (first
// kernel 15 -> 6 us).  Trips of 32 / 64 values in the block loops and a chain trip of 64 end values in the finishing kernel
// were measured slower or equal (tools/exp/README.md).
constexpr int kEnergyBatch = FRT_ENERGY_BATCH, kEnergyChain = FRT_ENERGY_CHAIN;
static_assert(kEnergySplit % kEnergyBatch == 0, "a split is walked in whole batches");

// seg_end: [channel][nsplit + 1][nbands]; slot nsplit of a channel holds the carry the batch starts from (a copy of `smooth`
// taken here, so that the finishing launch may replace `smooth` while its other workgroups still need the old value).
// FULL: the split is whole (every split but a batch's last may be ragged) — no bounds in the loops, and the compiler barrier keeps
// the next trip's requests in front of this trip's stores (left alone it sinks them behind the stores and waits for all of them).
template <bool FULL>
__device__ __forceinline__ void energy_local_body(double* p, double* __restrict__ end_slot, double d, int count, int nbands) {
    double local = 0.0;
    double nx[kEnergyBatch];
    auto request = [&](int i) {
#pragma unroll
        for (int j = 0; j < kEnergyBatch; ++j) nx[j] = (FULL || i + j < count) ? p[(size_t)(i + j) * nbands] : 0.0;
    };
    request(0);
    for (int i = 0; i < count; i += kEnergyBatch) {
        double e[kEnergyBatch];
#pragma unroll
        for (int j = 0; j < kEnergyBatch; ++j) e[j] = nx[j];
        if (i + kEnergyBatch < count) request(i + kEnergyBatch);      // (entries this trip does not write)
        if (FULL) asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < kEnergyBatch; ++j) {
            if (FULL || i + j < count) {
                local = e[j] + local * d;
                p[(size_t)(i + j) * nbands] = local;
            }
        }
    }
    *end_slot = local;
}

__global__ void __launch_bounds__(256) energy_local_kernel(double* eblock, const double* __restrict__ decay_n, double* __restrict__ seg_end,
                                                           const double* __restrict__ smooth, int nblocks, int nbands) {
    const int band = threadIdx.x, sp = blockIdx.x, c = blockIdx.y, nsplit = gridDim.x;
    if (band >= nbands) return;
    const int b0 = sp * kEnergySplit, b1 = (b0 + kEnergySplit) < nblocks ? (b0 + kEnergySplit) : nblocks;
    const double d = decay_n[band];
    double* p = eblock + ((size_t)c * nblocks + b0) * nbands + band;
    if (sp == 0) seg_end[((size_t)c * (nsplit + 1) + nsplit) * nbands + band] = smooth[(size_t)c * nbands + band];
    double* end_slot = seg_end + ((size_t)c * (nsplit + 1) + sp) * nbands + band;
    if (b1 - b0 == kEnergySplit) energy_local_body<true>(p, end_slot, d, kEnergySplit, nbands);
    else energy_local_body<false>(p, end_slot, d, b1 - b0, nbands);
}

struct EnergyOut {
    void* out;
    const double* weight_db;
    int sub, nbands;
    size_t row0;                 // index of the channel's first output row
};

// carry: the smoothed value in front of the split; returns the split's last smoothed value
template <bool FULL, bool AS_DB, bool OUT_F32>
__device__ __forceinline__ double energy_finish_body(const double* __restrict__ lp, double nx[kEnergyBatch], double carry, double d, int b0,
                                                     int count, int band, const EnergyOut& o) {
    const double w = (AS_DB && o.weight_db) ? o.weight_db[band] : 0.0;
    double pw = d, last = carry;
    // entry b is a caller's value when (b + 1) % sub == 0, its row b / sub: counted, not divided (two integer divisions per
    // entry were most of this kernel's instructions)
    int rem = b0 % o.sub;
    size_t idx = (o.row0 + (size_t)(b0 / o.sub)) * o.nbands + band;
    for (int i = 0; i < count; i += kEnergyBatch) {
        double e[kEnergyBatch];
#pragma unroll
        for (int j = 0; j < kEnergyBatch; ++j) e[j] = nx[j];
        if (i + kEnergyBatch < count) {
#pragma unroll
            for (int j = 0; j < kEnergyBatch; ++j) nx[j] = (FULL || i + kEnergyBatch + j < count) ? lp[(size_t)(i + kEnergyBatch + j) * o.nbands] : 0.0;
        }
        if (FULL) asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < kEnergyBatch; ++j) {
            if (FULL || i + j < count) {
                last = e[j] + carry * pw;
                pw *= d;
                if (++rem != o.sub) continue;                       // an entry inside a caller's block
                rem = 0;
                double v = last;
                if (AS_DB) v = 10.0 * log10(v + 1e-30) + w;
                if (OUT_F32) ((float*)o.out)[idx] = (float)v;
                else ((double*)o.out)[idx] = v;
                idx += o.nbands;
            }
        }
    }
    return last;
}

// AS_DB / OUT_F32 are instances, not branches: the conversion's code (a float64 log10 per entry) between the entries of the linear
// instance made every trip a string of taken branches through cold instruction-cache lines — most of the launch's 20 us.
template <bool AS_DB, bool OUT_F32>
__global__ void __launch_bounds__(256) energy_finish_kernel(const double* __restrict__ local, const double* __restrict__ decay_n,
                                                            const double* __restrict__ seg_end, double* __restrict__ smooth, void* __restrict__ out,
                                                            int nblocks, int nbands, const double* __restrict__ weight_db, int sub) {
    const int band = threadIdx.x, sp = blockIdx.x, c = blockIdx.y, nsplit = gridDim.x;
    if (band >= nbands) return;
    const int b0 = sp * kEnergySplit, b1 = (b0 + kEnergySplit) < nblocks ? (b0 + kEnergySplit) : nblocks;
    const double d = decay_n[band];
    const double* lp = local + ((size_t)c * nblocks + b0) * nbands + band;
    double nx[kEnergyBatch];                                     // the split's first values travel while the chain is formed
#pragma unroll
    for (int j = 0; j < kEnergyBatch; ++j) nx[j] = j < b1 - b0 ? lp[(size_t)j * nbands] : 0.0;
    double dlen = 1.0;                                           // d^kEnergySplit (every split before this one is full)
    for (int i = 0; i < kEnergySplit; ++i) dlen *= d;
    const double* se = seg_end + (size_t)c * (nsplit + 1) * nbands + band;
    double carry = se[(size_t)nsplit * nbands];                  // the smoothed value the batch starts from (energy_local_kernel's copy)
    for (int g = 0; g < sp; g += kEnergyChain) {
        double e[kEnergyChain];
#pragma unroll
        for (int j = 0; j < kEnergyChain; ++j) e[j] = g + j < sp ? se[(size_t)(g + j) * nbands] : 0.0;
#pragma unroll
        for (int j = 0; j < kEnergyChain; ++j)
            if (g + j < sp) carry = e[j] + carry * dlen;
    }
    const EnergyOut o{out, weight_db, sub, nbands, (size_t)c * (nblocks / sub)};
    const double last = (b1 - b0 == kEnergySplit) ? energy_finish_body<true, AS_DB, OUT_F32>(lp, nx, carry, d, b0, kEnergySplit, band, o)
                                                  : energy_finish_body<false, AS_DB, OUT_F32>(lp, nx, carry, d, b0, b1 - b0, band, o);
    if (sp == nsplit - 1) smooth[(size_t)c * nbands + band] = last;      // the other splits read energy_local_kernel's copy
}

// ---- host side -----------------------------------------------------------------------------------

// A^L for the DF2T state recurrence z' = A z (zero input): A[n][n+1] = 1, A[n][0] -= a[n+1].
static void transition_power(const double* a_coef, int order, long long L, double* out /*16x16*/) {
    const int d = kStates;
    std::vector<long double> A(d * d, 0.0L), R(d * d, 0.0L), T(d * d);
    for (int n = 0; n < order; ++n) {
        if (n + 1 < order) A[n * d + n + 1] = 1.0L;
        A[n * d + 0] -= (long double)a_coef[n + 1];
    }
    for (int n = 0; n < d; ++n) R[n * d + n] = 1.0L;
    auto mul = [&](std::vector<long double>& X, const std::vector<long double>& Y) {
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                long double acc = 0;
                for (int k = 0; k < d; ++k) acc += X[i * d + k] * Y[k * d + j];
                T[i * d + j] = acc;
            }
        X = T;
    };
    while (L > 0) {
        if (L & 1) mul(R, A);
        L >>= 1;
        if (L) {
            std::vector<long double> A2 = A;
            mul(A2, A);
            A = A2;
        }
    }
    for (int i = 0; i < d * d; ++i) out[i] = (double)R[i];
}

}  // namespace frt

using namespace frt;

extern "C" int64_t frt_octbank_packed_length(const frt_octbank* h, int n) {
    if (!h || n < 0) return 0;
    int len[kNOctave];
    stage_lengths(n, len);
    int64_t total = 0;
    for (int j = 0; j < kNOctave; ++j) total += (int64_t)len[j] * h->bpo;
    return total;
}

extern "C" int frt_octbank_state_length(const frt_octbank* h) { return h ? kNOctave * (4 * h->bpo + 12) : 0; }

extern "C" int frt_octbank_create(frt_octbank** out, int bands_per_octave, int n_channels, int mode, const double* boct,
                                  const double* aoct, const double* bdec, const double* adec, const double* boct_fir,
                                  const double* bdec_fir) {
    FRT_REQUIRE(out, "frt_octbank_create: null handle pointer");
    *out = nullptr;
    FRT_REQUIRE(bands_per_octave >= 0 && bands_per_octave <= 24, "frt_octbank_create: bands_per_octave %d not in [0, 24]",
                bands_per_octave);
    FRT_REQUIRE(n_channels >= 1, "frt_octbank_create: n_channels %d < 1", n_channels);
    FRT_REQUIRE(mode == 0 || mode == 1, "frt_octbank_create: mode %d unknown (0 = exact IIR, 1 = FFT overlap-add)", mode);
    FRT_REQUIRE(mode == 0 || (boct_fir && bdec_fir && bands_per_octave >= 1),
                "frt_octbank_create: mode 1 needs the FIR taps and at least one band per octave");
    FRT_REQUIRE(bdec && adec && (bands_per_octave == 0 || (boct && aoct)), "frt_octbank_create: null coefficients");
    frt_octbank* h = new frt_octbank();
    h->bpo = bands_per_octave;
    h->n_channels = n_channels;
    h->mode = mode;
    h->nbands = kNOctave * bands_per_octave;
    h->nfilt = bands_per_octave + 1;
    h->use_graph = exp_env("FRT_NO_GRAPH") == nullptr;
    h->h_coef.assign((size_t)h->nfilt * kCoefStride, 0.0);
    h->h_order.assign(h->nfilt, 0);
    for (int i = 0; i < bands_per_octave; ++i) {       // 4th-order band-passes, 5 + 5 coefficients
        for (int t = 0; t < 5; ++t) {
            h->h_coef[(size_t)i * kCoefStride + t] = boct[i * 5 + t];
            h->h_coef[(size_t)i * kCoefStride + kMaxOrder + 1 + t] = aoct[i * 5 + t];
        }
        h->h_order[i] = 4;
    }
    for (int t = 0; t < 13; ++t) {                       // 12th-order decimator, 13 + 13 coefficients
        h->h_coef[(size_t)bands_per_octave * kCoefStride + t] = bdec[t];
        h->h_coef[(size_t)bands_per_octave * kCoefStride + kMaxOrder + 1 + t] = adec[t];
    }
    h->h_order[bands_per_octave] = 12;
    int rc;
    if ((rc = upload(h->coef, h->h_coef)) || (rc = upload(h->order, h->h_order)) ||
        (rc = h->state.reserve(kNOctave * h->stage_state_elems() * sizeof(double)))) {
        frt_octbank_destroy(h);
        return rc;
    }
    if (hipMemset(h->state.ptr, 0, h->state.bytes) != hipSuccess) {
        set_last_error("frt_octbank_create: hipMemset failed");
        frt_octbank_destroy(h);
        return FRT_ERR_HIP;
    }
    if (mode == 1 && (rc = frt_ola_create(h, boct_fir, bdec_fir))) {
        frt_octbank_destroy(h);
        return rc;
    }
    *out = h;
    return FRT_OK;
}

extern "C" void frt_octbank_destroy(frt_octbank* h) {
    if (!h) return;
    free_retired_allocations(true);      // blocks parked by growing buffers (common.h); synchronises the device like the releases below
    for (auto& e : h->graphs)
        if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (h->gstream) (void)hipStreamDestroy(h->gstream);
    for (auto& st : h->side)
        if (st) (void)hipStreamDestroy(st);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    for (auto& e : h->ev_x)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : h->ev_side)
        if (e) (void)hipEventDestroy(e);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    frt_ola_destroy(h);
    DeviceBuffer* bufs[] = {&h->coef, &h->order, &h->state, &h->state_snap, &h->xin, &h->ypacked, &h->chunk_end, &h->chunk_init, &h->power, &h->zs_table, &h->zs_table_m, &h->zs_rowmap, &h->eseg,
                            &h->eblock, &h->alpha, &h->decay_n, &h->smooth, &h->weight, &h->eout};
    for (auto* b : bufs) b->release();
    for (auto& b : h->xbuf) b.release();
    delete h;
}

extern "C" int frt_octbank_set_stream(frt_octbank* h, void* s) {
    FRT_REQUIRE(h, "frt_octbank_set_stream: null handle");
    h->stream = (hipStream_t)s;
    return FRT_OK;
}

extern "C" int frt_octbank_set_chunk(frt_octbank* h, int chunk0) {
    FRT_REQUIRE(h, "frt_octbank_set_chunk: null handle");
    // A/B and tests: a negative value selects the same chunking with pass 1 run as a second recurrence
    h->zero_state_by_recurrence = chunk0 < 0;
    if (chunk0 < 0) chunk0 = -chunk0;
    FRT_REQUIRE(chunk0 == 0 || (chunk0 >= 256 && chunk0 % 64 == 0),
                "frt_octbank_set_chunk: chunk %d must be 0 (sequential) or a multiple of 64, at least 256", chunk0);
    h->chunk0 = chunk0;
    return FRT_OK;
}

extern "C" int frt_octbank_reset(frt_octbank* h) {
    FRT_REQUIRE(h, "frt_octbank_reset: null handle");
    if (h->gstream) FRT_HIP_CHECK(hipStreamSynchronize(h->gstream));
    FRT_HIP_CHECK(hipMemsetAsync(h->state.ptr, 0, h->state.bytes, h->stream));
    if (h->ola) {
        int rc = frt_ola_reset(h);
        if (rc) return rc;
    }
    if (h->smooth.ptr) FRT_HIP_CHECK(hipMemsetAsync(h->smooth.ptr, 0, h->smooth.bytes, h->stream));
    return FRT_OK;
}

// State exchange in the reference's order (filter.py:121-133): per channel, per octave j the band
// states i = bpo-1 .. 0 (4 doubles each) followed by the decimator's 12.
static int state_copy(frt_octbank* h, double* user, bool to_user) {
    const size_t per_stage = h->stage_state_elems();
    std::vector<double> dev(kNOctave * per_stage);
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (to_user) FRT_HIP_CHECK(hipMemcpy(dev.data(), h->state.ptr, dev.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int slen = frt_octbank_state_length(h);
    for (int c = 0; c < h->n_channels; ++c) {
        double* u = user + (size_t)c * slen;
        int pos = 0;
        for (int j = 0; j < kNOctave; ++j) {
            for (int i = h->bpo - 1; i >= -1; --i) {
                const int f = i >= 0 ? i : h->bpo;
                const int ord = h->h_order[f];
                double* d = &dev[j * per_stage + ((size_t)c * h->nfilt + f) * kStates];
                for (int s = 0; s < ord; ++s) {
                    if (to_user) u[pos + s] = d[s];
                    else d[s] = u[pos + s];
                }
                pos += ord;
            }
        }
    }
    if (!to_user) FRT_HIP_CHECK(hipMemcpy(h->state.ptr, dev.data(), dev.size() * sizeof(double), hipMemcpyHostToDevice));
    return FRT_OK;
}

extern "C" int frt_octbank_get_state(frt_octbank* h, double* z) {
    FRT_REQUIRE(h && z, "frt_octbank_get_state: null argument");
    return state_copy(h, z, true);
}

extern "C" int frt_octbank_set_state(frt_octbank* h, const double* z) {
    FRT_REQUIRE(h && z, "frt_octbank_set_state: null argument");
    std::vector<double> zero(kNOctave * h->stage_state_elems(), 0.0);
    FRT_HIP_CHECK(hipMemcpy(h->state.ptr, zero.data(), zero.size() * sizeof(double), hipMemcpyHostToDevice));
    return state_copy(h, const_cast<double*>(z), false);
}

// Samples per time chunk at octave stage j: the stage-0 chunk scaled to the stage's rate, a multiple of 64
// (one wavefront load of samples), never below 64.
static int stage_chunk(int chunk0, int j) {
    int c = (chunk0 >> j) / 64 * 64;
    return c < 64 ? 64 : c;
}

// Chunks per scan row of a stage (iir_scan_kernel): the smallest power of two g for which M = A^(L g) satisfies
// max |M^2| < 1e-20 for every filter of the bank, so that a row's true initial state needs its two predecessors only.
static int scan_group_for(const frt_octbank* h, int L, int nchunks) {
    int g = 1;
    std::vector<double> M(kStates * kStates);
    for (; g < nchunks; g *= 2) {
        double worst = 0.0;
        for (int f = 0; f < h->nfilt; ++f) {
            const double* ac = &h->h_coef[(size_t)f * kCoefStride + kMaxOrder + 1];
            transition_power(ac, h->h_order[f], (long long)L * g, M.data());
            for (int i = 0; i < kStates; ++i)
                for (int j = 0; j < kStates; ++j) {
                    double acc = 0.0;
                    for (int k = 0; k < kStates; ++k) acc += M[i * kStates + k] * M[k * kStates + j];
                    worst = std::fmax(worst, std::fabs(acc));
                }
        }
        if (worst < 1e-20) break;
    }
    return g < nchunks ? g : (nchunks > 0 ? nchunks : 1);
}

// A^L and A^(L group) of every (stage, filter) for the current chunking of n input samples.
static int ensure_powers(frt_octbank* h, int n) {
    if (h->power_chunk0 == h->chunk0 && h->power_n == n) return FRT_OK;
    int len[kNOctave];
    stage_lengths(n, len);
    const size_t per = (size_t)kNOctave * h->nfilt * kStates * kStates;
    std::vector<double> p(2 * per);
    h->sgroup.assign(kNOctave, 1);
    h->shalo.assign(kNOctave, 2);
    h->slook.assign(kNOctave, 65);
    for (int j = 0; j < kNOctave; ++j) {
        const int cj = stage_chunk(h->chunk0, j);
        const int nj = (len[j] + cj - 1) / cj;
        // g chunks make M = A^(L g) with M^2 = 0: the decay spans 2 g chunks = `halo` rows of sgroup chunks
        const int g = scan_group_for(h, cj, nj);
        int rg = kScanRowMax;
        while ((2 * g + rg - 1) / rg > kScanRows / 2) rg *= 2;              // at most half of a workgroup's rows are halo
        // chunks the decay spans, for the look-back form of the output pass: the smallest K with max |A^(L K)| < 1e-20 over the filters
        {
            std::vector<double> MK(kStates * kStates);
            int K = 1;
            for (; K <= 64; ++K) {
                double worst = 0.0;
                for (int f = 0; f < h->nfilt; ++f) {
                    transition_power(&h->h_coef[(size_t)f * kCoefStride + kMaxOrder + 1], h->h_order[f], (long long)cj * K, MK.data());
                    for (double v : MK) worst = std::fmax(worst, std::fabs(v));
                }
                if (worst < 1e-20) break;
            }
            h->slook[j] = K;
        }
        h->sgroup[j] = g <= rg || exp_env("FRT_IIR_LONG_SCAN_ROWS") ? g : rg;
        h->shalo[j] = g <= rg || exp_env("FRT_IIR_LONG_SCAN_ROWS") ? 2 : (2 * g + rg - 1) / rg;
        for (int f = 0; f < h->nfilt; ++f) {
            const double* ac = &h->h_coef[(size_t)f * kCoefStride + kMaxOrder + 1];
            transition_power(ac, h->h_order[f], cj, &p[((size_t)j * h->nfilt + f) * kStates * kStates]);
            transition_power(ac, h->h_order[f], (long long)cj * h->sgroup[j], &p[per + ((size_t)j * h->nfilt + f) * kStates * kStates]);
        }
    }
    int rc = upload(h->power, p);
    if (rc) return rc;
    // zero-state response tables g[k][row] = (A^(L-1-k) B)[s], rows = the live states of every filter.  The decimator's rows come
    // first, padded to whole tiles of 16 rows, then the band filters': a launch of the table product can serve the decimator alone
    // (the only filter the next stage waits for) or the band filters alone (ZeroStateArgs::rt_base / rt_count).
    std::vector<int> rowmap, row_of(h->nfilt, 0);
    const int fdec = h->nfilt - 1;                  // the decimator is the last filter of every stage (frt_octbank_create)
    for (int t = 0; t < h->h_order[fdec]; ++t) rowmap.push_back(fdec * kStates + t);
    rowmap.resize((rowmap.size() + 15) / 16 * 16, -1);
    h->zs_dec_tiles = (int)rowmap.size() / 16;
    for (int f = 0; f < fdec; ++f) {
        row_of[f] = (int)rowmap.size();
        for (int t = 0; t < h->h_order[f]; ++t) rowmap.push_back(f * kStates + t);
    }
    h->zs_rows = (int)rowmap.size();
    rowmap.resize((rowmap.size() + 15) / 16 * 16, -1);
    h->zs_band_tiles = (int)rowmap.size() / 16 - h->zs_dec_tiles;
    h->zs_rows_padded = (int)rowmap.size();
    h->zs_offset.assign(kNOctave, 0);
    size_t total = 0;
    for (int j = 0; j < kNOctave; ++j) {
        h->zs_offset[j] = total;
        total += (size_t)stage_chunk(h->chunk0, j) * h->zs_rows_padded;
    }
    std::vector<double> tab(total, 0.0), tab_m(total, 0.0);      // tab_m: the same values in the MFMA kernel's operand order
    for (int j = 0; j < kNOctave; ++j) {
        const int L = stage_chunk(h->chunk0, j);
        for (int f = 0; f < h->nfilt; ++f) {
            const int ord = h->h_order[f], row = row_of[f];
            const double* bc = &h->h_coef[(size_t)f * kCoefStride];
            const double* ac = bc + kMaxOrder + 1;
            std::vector<long double> v(ord), w(ord);
            for (int t = 0; t < ord; ++t) v[t] = (long double)bc[t + 1] - (long double)ac[t + 1] * (long double)bc[0];
            for (int k = L - 1; k >= 0; --k) {
                for (int t = 0; t < ord; ++t) {
                    tab[h->zs_offset[j] + (size_t)k * h->zs_rows_padded + row + t] = (double)v[t];
                    tab_m[h->zs_offset[j] + zs_mfma_index(k, row + t, h->zs_rows_padded / 16)] = (double)v[t];
                }
                // v <- A v:  (A v)[t] = v[t+1] - a[t+1] v[0]
                for (int t = 0; t < ord; ++t) w[t] = (t + 1 < ord ? v[t + 1] : 0.0L) - (long double)ac[t + 1] * v[0];
                v.swap(w);
            }
        }
    }
    if ((rc = upload(h->zs_table, tab)) || (rc = upload(h->zs_table_m, tab_m)) || (rc = upload(h->zs_rowmap, rowmap))) return rc;
    h->power_chunk0 = h->chunk0;
    h->power_n = n;
    return FRT_OK;
}

// Runs the nine octave stages on device buffers.  d_x: stage-0 input; d_y (nullable): packed band
// outputs; energies (nullable eblock): zero-state block energies for blocks of `eblock0` input samples.
// `esub`: every esub-th entry of the block axis is one the caller reads (frt_octbank_energies: 1 unless the time-parallel chunk
// is shorter than the caller's block).
static int run_stages(frt_octbank* h, const void* d_x, int in_f32, long long x_stride, int n, double* d_y, int64_t y_cstride,
                      double* d_eblock, int eblock0, int nblocks, int esub = 1) {
    int len[kNOctave];
    stage_lengths(n, len);
    const bool parallel = h->chunk0 > 0 && n >= 2 * h->chunk0;
    // K-slices of a stage's table product: enough wavefronts to fill the chip when the columns (channel x chunk) alone do not
    static const bool use_vector_alu = exp_env("FRT_ZS_VECTOR") != nullptr;      // A/B runs: the vector-ALU kernel
    auto slices_for = [&](int chunk, int nch) {
        const int cols_per_wave = use_vector_alu ? 64 * kZsCols : 16;
        const int slice_min = use_vector_alu ? 8 : 16;
        const long long colwaves = ((long long)h->n_channels * nch + cols_per_wave - 1) / cols_per_wave;
        static const int max_slices = exp_int("FRT_ZS_MAX_SLICES", kMaxSlices);      // A/B runs
        // (round 4: one wavefront per CU is where splitting K stops paying — every doubling adds the partial sums'
        // traffic and, from 2 slices on, a launch that sums them: 2048 -> 256 took configs[4]'s 216-band bank from
        // 0.90 to 0.72 ms and configs[2]'s from 0.72 to 0.70, profiles/r04_zero_state_slices.txt)
        static const int wave_goal = exp_int("FRT_ZS_WAVE_GOAL", device_cu_count());
        // (a slice is at least 64 samples: below that the product is launch-bound whatever its grid, and the launch that
        // sums the slices costs its ~5 us — the 64-sample chunks of the low-rate stages are not sliced: -1 % at 8 ch x 27 bands)
        int ns = 1;
        while (ns < max_slices && colwaves * ns < wave_goal && chunk % (2 * ns * slice_min) == 0 && chunk / (2 * ns) >= 64) ns *= 2;      // whole K-blocks per slice
        return ns;
    };
    // every stage has its own scratch (zero-state end states per K-slice, chunk start states): stages overlap in time (below)
    size_t off_end[kNOctave] = {}, off_init[kNOctave] = {};
    int stage_slices[kNOctave] = {};
    if (parallel) {
        size_t tot_end = 0, tot_init = 0;
        for (int j = 0; j < kNOctave; ++j) {
            int cj = stage_chunk(h->chunk0, j);
            if (cj < 64) cj = 64;
            const int nj = (len[j] + cj - 1) / cj;
            const size_t ws = (size_t)h->n_channels * h->nfilt * (nj > 0 ? nj : 1) * kStates;
            stage_slices[j] = slices_for(cj, nj);
            off_end[j] = tot_end;
            off_init[j] = tot_init;
            tot_end += ws * stage_slices[j];
            tot_init += ws;
        }
        int rc = ensure_powers(h, n);
        if (rc) return rc;
        if ((rc = h->chunk_end.reserve(tot_end * sizeof(double))) || (rc = h->chunk_init.reserve(tot_init * sizeof(double))))
            return rc;
    }
    // EXPERIMENT (-DFRT_EXPERIMENTS builds, FRT_IIR_SIDE=1; measured and rejected, profiles/r05_iir_side_streams.txt): the band
    // filters of a stage beside the decimator chain.  Only a stage's DECIMATOR feeds the next stage, so its table product, chunk
    // scan and output pass run on the handle's stream one stage after the other, and every stage's band filters — the same three
    // launches restricted to them — follow on a side stream as soon as the stage's input exists.  It does not pay: the
    // decimator-only launches last what the all-filter launches last (the table product is bound by its sample loads, the output
    // pass by its dependent float64 chain, the low-rate stages by launch latency), so the chain is no shorter and the band
    // launches beside it slow it down: 8 ch x 27 bands 0.64 -> 0.81 ms, 8 ch x 216 bands 0.69 -> 0.72, 64 ch x 216 bands 4.62 -> 4.47.
    bool beside = parallel && d_y == nullptr && d_eblock != nullptr && h->bpo >= 1 && !h->zero_state_by_recurrence && !use_vector_alu &&
                  !CaptureScope::active() && exp_env("FRT_IIR_SIDE") != nullptr;
    if (beside) {
        if (!h->ev_start) {
            FRT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
            for (auto& e : h->ev_x) FRT_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& e : h->ev_side) FRT_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& st : h->side) FRT_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        }
        FRT_HIP_CHECK(hipEventRecord(h->ev_start, h->stream));
    }
    bool side_used[frt_octbank::kSideStreams] = {};
    for (int j = 1; j < kNOctave; ++j) {
        int rc = h->xbuf[j].reserve((size_t)h->n_channels * len[j] * sizeof(double));
        if (rc) return rc;
    }
    // packed offsets: band k (low band first) has length len[j], j = 8 - k / bpo
    std::vector<long long> band_off(h->nbands + 1, 0);
    for (int k = 0; k < h->nbands; ++k) band_off[k + 1] = band_off[k] + len[kNOctave - 1 - k / h->bpo];

    for (int j = 0; j < kNOctave; ++j) {
        IirStageArgs a{};
        a.x = j == 0 ? d_x : h->xbuf[j].ptr;
        a.x_stride = j == 0 ? x_stride : len[j];
        a.n = len[j];
        a.in_f32 = j == 0 ? in_f32 : 0;
        a.coef = h->coef.as<double>();
        a.order = h->order.as<int>();
        a.nfilt = h->nfilt;
        a.dec_filter = h->bpo;
        a.state = h->state.as<double>() + (size_t)j * h->stage_state_elems();
        a.chunk = parallel ? stage_chunk(h->chunk0, j) : ((len[j] + 63) / 64 * 64);
        if (a.chunk < 64) a.chunk = 64;
        a.nchunks = parallel ? (len[j] + a.chunk - 1) / a.chunk : 1;
        a.chunk_end = h->chunk_end.as<double>() + off_end[j];
        a.chunk_init = h->chunk_init.as<double>() + off_init[j];
        a.scan_group = parallel ? h->sgroup[j] : 1;
        a.scan_rows = (a.nchunks + a.scan_group - 1) / a.scan_group;
        a.y = d_y;
        a.y_cstride = y_cstride;
        for (int i = 0; i < h->bpo; ++i) {
            const int k = (kNOctave - 1 - j) * h->bpo + i;
            a.y_off[i] = band_off[k];
            a.band_index[i] = k;
        }
        a.y_off[h->bpo] = 0;
        a.band_index[h->bpo] = -1;
        // the reference also runs the last octave's decimator (its state is carried), output unused
        a.xnext = j + 1 < kNOctave ? h->xbuf[j + 1].as<double>() : nullptr;
        a.xnext_stride = j + 1 < kNOctave ? len[j + 1] : 0;
        a.eblock = d_eblock;
        // An energy block of this stage is eblock0 / 2^j of its samples: 1 or 2 at the lowest rates when the block axis is finer
        // than 1024 samples.  The lane kernel works in groups of 4 samples: it may take a block of 4 that spans `mul` entries of
        // the block axis (zeros in the first mul - 1, the group's energy in the last — the smoothing recurrence then lands on
        // the right value at every mul-th entry) ONLY when every entry the caller reads is such a last one, i.e. mul divides
        // esub.  Everything else — the sequential mode, the slot kernel, a caller that reads every entry — keeps the true
        // block length (the slot kernel's short-block loop serves 1 and 2).
        auto set_eblock = [&](int len, int mul) {
            a.eblock_len = len;
            a.eblock_mul = mul;
            a.eblock_shift = 0;
            while ((1 << a.eblock_shift) < a.eblock_len) ++a.eblock_shift;
        };
        const int elen_true = d_eblock ? (eblock0 >> j) : 1;
        FRT_REQUIRE(elen_true >= 1, "octave bank: energy block of %d samples is shorter than one sample of stage %d", eblock0, j);
        set_eblock(elen_true, 1);
        if (d_eblock && parallel && elen_true < 4 && esub % (4 / elen_true) == 0) set_eblock(4, 4 / elen_true);
        a.nblocks = nblocks;
        a.nbands = h->nbands;
        a.alpha = h->alpha.as<double>();
        if (a.n == 0) continue;
        int rc;
        if (!parallel) {
            a.pass = 0;
            if ((rc = launch_iir_stage(a, h->h_order.data(), h->n_channels, h->stream))) return rc;
        } else {
            const int n_slices = stage_slices[j];
            const long long slice_stride = (long long)h->n_channels * h->nfilt * a.nchunks * kStates;
            const int fdec = h->nfilt - 1;
            a.pass = 2;
            static const bool exact_ops = exp_env("FRT_IIR_EXACT_OPS") != nullptr;      // A/B runs
            a.fused = (d_y == nullptr && d_eblock != nullptr && !exact_ops) ? 1 : 0;
            a.n_channels = h->n_channels;
            const bool lane_serves = lane_kernel_serves(a, h->h_order.data());
            // The look-back form of the output pass (no scan launch) while the decay spans few chunks.  Measured (profiles/r06_iir_lookback.txt,
            // 8 ch x 27 bands, chunks of 1024): stage 0 (K = 6) output pass 135 -> 142 us for a scan launch of 19 us less; stage 1 (K = 13)
            // 65 -> 88 us for 15 us less — a Horner step costs a lane 1.4-1.9 us (192 multiply-adds fed through the scalar cache) against
            // 0.3 us in the scan's DPP rows: it paid up to K = 8 only.  Later in round 6 (profiles/r06_iir_lookback.txt, second part): most of
            // a step's time was a memory round trip — every step's request of the next end state was awaited where it was issued — and
            // the product walked one dependent chain per row; with the request pinned in front of a column-by-column product a step is
            // ~0.9 us: stage 0 (K = 6) 16 + 109 -> 116 us, stage 1 (K = 13) 12 + 62 -> 73 us (even), 216 bands' stage 0 (K = 12) 19 + 115
            // -> 127 us; K = 25 still loses (18 + 36 -> 60 us).  Up to K = 16.
            static const int look_max = exp_int("FRT_IIR_LOOKBACK_MAX", kLookbackMax);
            const bool look = lane_serves && !beside && !h->zero_state_by_recurrence && !use_vector_alu && h->slook[j] <= look_max && option(kOptIirLookback) != 0 &&
                              !lane_split_serves(a, h->n_channels, kWhichAll);
            if (look) {
                if ((rc = h->state_snap.reserve((size_t)kNOctave * h->stage_state_elems() * sizeof(double)))) return rc;
                a.lookback = h->slook[j];
                a.power_l = h->power.as<double>() + (size_t)j * h->nfilt * kStates * kStates;
                a.state_in = h->state_snap.as<double>() + (size_t)j * h->stage_state_elems();
            }
            ZeroStateArgs z{};
            z.x = a.x; z.x_stride = a.x_stride; z.n = a.n; z.in_f32 = a.in_f32;
            z.chunk = a.chunk; z.nchunks = a.nchunks; z.n_channels = h->n_channels; z.nfilt = h->nfilt;
            z.slice = a.chunk / n_slices;
            z.vec = ((uintptr_t)a.x % 16 == 0) && (a.x_stride % (a.in_f32 ? 4 : 2) == 0);
            z.table = h->zs_table.as<double>() + h->zs_offset[j];
            z.table_m = h->zs_table_m.as<double>() + h->zs_offset[j];
            z.rows = h->zs_rows; z.rows_padded = h->zs_rows_padded;
            z.rowmap = h->zs_rowmap.as<int>();
            z.order = h->order.as<int>();
            double* const cend = h->chunk_end.as<double>() + off_end[j];
            double* const cinit = h->chunk_init.as<double>() + off_init[j];
            z.partial = cend;
            z.partial_stride = slice_stride;
            if (look) {
                z.snap_src = a.state;
                z.snap_dst = h->state_snap.as<double>() + (size_t)j * h->stage_state_elems();
                z.snap_count = (int)h->stage_state_elems();
            }
            const long long colwaves = ((long long)h->n_channels * a.nchunks + 15) / 16;
            const size_t per = (size_t)kNOctave * h->nfilt * kStates * kStates, off = (size_t)j * h->nfilt * kStates * kStates;
            const int halo = h->shalo[j], nseg = (a.scan_rows + (kScanRows - halo) - 1) / (kScanRows - halo);
            // table product, slice sum and chunk scan of the filters f0 .. f0 + nf - 1 (row tiles t0 .. t0 + nt - 1) on `st`
            auto front = [&](hipStream_t st, int t0, int nt, int f0, int nf) {
                z.rt_base = t0;
                z.rt_count = nt;
                // two row tiles per workgroup (every sample feeds two MFMAs) unless the range is a single tile; an odd range's last
                // workgroup computes one tile it does not store (216 bands: 7 tiles, 4 workgroups along z as in round 4)
                if (nt >= 2)
                    hipLaunchKernelGGL(iir_zero_state_mfma_kernel<2>, dim3((unsigned)colwaves, n_slices, (nt + 1) / 2), dim3(64), 0, st, z);
                else
                    hipLaunchKernelGGL(iir_zero_state_mfma_kernel<1>, dim3((unsigned)colwaves, n_slices, nt), dim3(64), 0, st, z);
                if (n_slices > 1) {
                    const long long cnt = (long long)a.nchunks * kStates * nf;
                    hipLaunchKernelGGL(iir_slice_sum_range_kernel, dim3((unsigned)((cnt + 255) / 256), h->n_channels), dim3(256), 0, st, cend,
                                       slice_stride, n_slices, h->nfilt, a.nchunks, f0, nf);
                }
                if (look) return;
                hipLaunchKernelGGL(iir_scan_kernel, dim3((unsigned)(h->n_channels * nf * nseg)), dim3(kScanRows * 16), 0, st,
                                   h->power.as<double>() + off, h->power.as<double>() + per + off, a.state, cend,
                                   h->order.as<int>(), cinit, h->nfilt, a.nchunks, a.scan_group, a.scan_rows, nseg, halo, f0, nf);
            };
            if (h->zero_state_by_recurrence || use_vector_alu) {
                // A/B paths of the table product: a second run of the recurrence, or the vector-ALU kernel; every filter in one chain
                if (h->zero_state_by_recurrence) {
                    a.pass = 1;
                    if ((rc = launch_iir_stage(a, h->h_order.data(), h->n_channels, h->stream))) return rc;
                    a.pass = 2;
                } else {
                    hipLaunchKernelGGL((iir_zero_state_kernel<kZsRowsPerPass, kZsCols>),
                                       dim3((unsigned)(((long long)h->n_channels * a.nchunks + 64 * kZsCols - 1) / (64 * kZsCols)), n_slices,
                                            h->zs_rows_padded / kZsRowsPerPass), dim3(64), 0, h->stream, z);
                    if (n_slices > 1)
                        hipLaunchKernelGGL(iir_slice_sum_kernel, dim3((unsigned)((slice_stride + 255) / 256)), dim3(256), 0, h->stream,
                                           cend, slice_stride, n_slices, slice_stride);
                }
                hipLaunchKernelGGL(iir_scan_kernel, dim3((unsigned)(h->n_channels * h->nfilt * nseg)), dim3(kScanRows * 16), 0, h->stream,
                                   h->power.as<double>() + off, h->power.as<double>() + per + off, a.state, cend,
                                   h->order.as<int>(), cinit, h->nfilt, a.nchunks, a.scan_group, a.scan_rows, nseg, halo, 0, h->nfilt);
                if (lane_serves) rc = launch_iir_lane(a, h->n_channels, h->stream, kWhichAll);
                else {
                    set_eblock(elen_true, 1);            // the slot kernel writes every entry of the block axis itself
                    rc = launch_iir_stage(a, h->h_order.data(), h->n_channels, h->stream);
                }
            } else if (beside && lane_serves) {
                // the decimator on the handle's stream ...
                front(h->stream, 0, h->zs_dec_tiles, fdec, 1);
                if ((rc = launch_iir_lane(a, h->n_channels, h->stream, kWhichDec))) return rc;
                FRT_HIP_CHECK(hipEventRecord(h->ev_x[j + 1], h->stream));
                // ... the band filters beside it, as soon as the stage's input exists
                const int sidx = j % frt_octbank::kSideStreams;
                hipStream_t sst = h->side[sidx];
                FRT_HIP_CHECK(hipStreamWaitEvent(sst, j == 0 ? h->ev_start : h->ev_x[j], 0));
                front(sst, h->zs_dec_tiles, h->zs_band_tiles, 0, fdec);
                rc = launch_iir_lane(a, h->n_channels, sst, kWhichBands);
                side_used[sidx] = true;
            } else {
                front(h->stream, 0, h->zs_dec_tiles + h->zs_band_tiles, 0, h->nfilt);
                if (lane_serves) rc = launch_iir_lane(a, h->n_channels, h->stream, kWhichAll);
                else {
                    set_eblock(elen_true, 1);            // the slot kernel writes every entry of the block axis itself
                    rc = launch_iir_stage(a, h->h_order.data(), h->n_channels, h->stream);
                }
                if (beside) FRT_HIP_CHECK(hipEventRecord(h->ev_x[j + 1], h->stream));      // (a later stage may still go beside)
            }
            if (rc) return rc;
            a.fused = 0;
        }
        FRT_HIP_CHECK(hipGetLastError());
    }
    for (int i = 0; i < frt_octbank::kSideStreams; ++i)
        if (side_used[i]) {                      // whatever follows on the handle's stream (the energy recurrences) waits for the band filters
            FRT_HIP_CHECK(hipEventRecord(h->ev_side[i], h->side[i]));
            FRT_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_side[i], 0));
        }
    return FRT_OK;
}


// Host-buffer path of frt_octbank_filter.  The first call with a block length runs eagerly (and
// allocates every device buffer); the second captures H2D copy + stage kernels + D2H copy into a
// hipGraph on a private stream; later calls replay it.  A graph is dropped when one of the device
// buffers it baked in has been re-allocated since.
static void snapshot_ptrs(const frt_octbank* h, const void** p) {
    p[0] = h->xin.ptr;
    p[1] = h->ypacked.ptr;
    for (int j = 1; j < kNOctave; ++j) p[1 + j] = h->xbuf[j].ptr;
    p[1 + kNOctave] = h->ola ? h->ola->pending.ptr : nullptr;
}

static int enqueue_filter(frt_octbank* h, int n, int64_t plen, hipStream_t s) {
    const size_t in_bytes = (size_t)h->n_channels * n * sizeof(double), out_bytes = (size_t)h->n_channels * plen * sizeof(double);
    FRT_HIP_CHECK(hipMemcpyAsync(h->xin.ptr, h->pin_in, in_bytes, hipMemcpyHostToDevice, s));
    hipStream_t keep = h->stream;
    h->stream = s;
    int rc = h->mode == 1 ? frt_ola_filter(h, h->xin.as<double>(), n, h->ypacked.as<double>(), plen)
                          : run_stages(h, h->xin.ptr, 0, n, n, h->ypacked.as<double>(), plen, nullptr, 0, 0);
    h->stream = keep;
    if (rc) return rc;
    FRT_HIP_CHECK(hipMemcpyAsync(h->pin_out, h->ypacked.ptr, out_bytes, hipMemcpyDeviceToHost, s));
    return FRT_OK;
}

static int filter_host(frt_octbank* h, const double* x, int n, double* y_packed, int64_t plen) {
    const size_t in_bytes = (size_t)h->n_channels * n * sizeof(double), out_bytes = (size_t)h->n_channels * plen * sizeof(double);
    int rc;
    if ((rc = h->xin.reserve(in_bytes)) || (rc = h->ypacked.reserve(out_bytes))) return rc;
    if (in_bytes > h->pin_in_bytes) {
        if (h->pin_in) (void)hipHostFree(h->pin_in);
        FRT_HIP_CHECK(hipHostMalloc(&h->pin_in, in_bytes, hipHostMallocDefault));
        h->pin_in_bytes = in_bytes;
    }
    if (out_bytes > h->pin_out_bytes) {
        if (h->pin_out) (void)hipHostFree(h->pin_out);
        FRT_HIP_CHECK(hipHostMalloc(&h->pin_out, out_bytes, hipHostMallocDefault));
        h->pin_out_bytes = out_bytes;
    }
    if (!h->gstream) FRT_HIP_CHECK(hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking));
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));        // order after earlier work on the caller's stream
    memcpy(h->pin_in, x, in_bytes);
    const bool no_chunk = option(kOptOlaChunkKernels) == 0;       // tests: the transform path on a call the chunk kernels would serve
    if (h->mode == 1 && n <= 1024 && in_bytes <= kZeroCopyMax && out_bytes <= kZeroCopyMax && !no_chunk) {
        // the production bank's block (Octave_Filters.filter): running convolutions, two launches, samples read and band
        // signals written in place in the page-locked blocks (ola.hip, chunk path) instead of nine transform launches
        if ((rc = frt_ola_chunk_filter(h, (const double*)h->pin_in, n, (double*)h->pin_out, plen))) return rc;
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        memcpy(y_packed, h->pin_out, out_bytes);
        return FRT_OK;
    }

    frt_octbank::StreamGraph* g = nullptr;
    for (auto& e : h->graphs)
        if (e.n == n) g = &e;
    const void* now[2 + kNOctave];
    const bool time_parallel = h->mode == 0 && h->chunk0 > 0 && n >= 2 * h->chunk0;   // allocates scratch per call shape
    if (g) {
        snapshot_ptrs(h, now);
        if (memcmp(now, g->ptrs, sizeof(now)) != 0) {         // a buffer moved: the graph is stale
            (void)hipGraphExecDestroy(g->exec);
            g->exec = nullptr;
            g->n = -1;
            g = nullptr;
            h->warmed_n = -1;
        }
    }
    if (g) {
        FRT_HIP_CHECK(hipGraphLaunch(g->exec, h->gstream));
    } else if (!h->use_graph || time_parallel || h->warmed_n != n) {
        if ((rc = enqueue_filter(h, n, plen, h->gstream))) return rc;
        h->warmed_n = n;
    } else {
        hipGraph_t graph = nullptr;
        FRT_HIP_CHECK(hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal));
        hipError_t e;
        {
            CaptureScope capturing;              // a buffer that grows in here parks its old block instead of freeing it
            rc = enqueue_filter(h, n, plen, h->gstream);
            e = hipStreamEndCapture(h->gstream, &graph);
        }
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        FRT_HIP_CHECK(e);
        frt_octbank::StreamGraph entry;
        entry.n = n;
        e = hipGraphInstantiate(&entry.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        FRT_HIP_CHECK(e);
        snapshot_ptrs(h, entry.ptrs);
        h->graphs.push_back(entry);
        FRT_HIP_CHECK(hipGraphLaunch(entry.exec, h->gstream));
    }
    FRT_HIP_CHECK(hipStreamSynchronize(h->gstream));
    memcpy(y_packed, h->pin_out, out_bytes);
    return FRT_OK;
}

extern "C" int frt_octbank_filter(frt_octbank* h, const double* x, int n, double* y_packed, int* dec_out) {
    FRT_REQUIRE(h && h->bpo >= 1, "frt_octbank_filter: needs a handle with bands");
    FRT_REQUIRE(n >= 0, "frt_octbank_filter: n %d < 0", n);
    if (dec_out)      // band k: dec = 2^(8 - k / bpo)  (octavefilters.py:60-63)
        for (int k = 0; k < h->nbands; ++k) dec_out[k] = 1 << (kNOctave - 1 - k / h->bpo);
    if (n == 0) {
        set_last_error("Filter input is too small");      // decimate.py:33-34
        return FRT_ERR_TOO_SMALL;
    }
    FRT_REQUIRE(x && y_packed, "frt_octbank_filter: null buffer");
    const int64_t plen = frt_octbank_packed_length(h, n);
    const bool dx = is_device_pointer(x), dy = is_device_pointer(y_packed);
    FRT_REQUIRE(dx == dy, "frt_octbank_filter: input and output must both be host or both be device memory");
    // mode 1: up to 1024 samples is the reference's own call (one overlap-add block, its FFT sizes); a longer input is
    // processed AS IF fed in 1024-sample blocks (the reference itself would crop it to its first stage's FFT size)
    if (h->mode == 1 && n > 1024) {
        if (dx) return frt_ola_filter_batch(h, x, 0, n, y_packed, plen, nullptr, 0, 0, nullptr);
        int rc;
        const size_t in_bytes = (size_t)h->n_channels * n * sizeof(double), out_bytes = (size_t)h->n_channels * plen * sizeof(double);
        if ((rc = h->xin.reserve(in_bytes)) || (rc = h->ypacked.reserve(out_bytes))) return rc;
        FRT_HIP_CHECK(hipMemcpyAsync(h->xin.ptr, x, in_bytes, hipMemcpyHostToDevice, h->stream));
        if ((rc = frt_ola_filter_batch(h, h->xin.ptr, 0, n, h->ypacked.as<double>(), plen, nullptr, 0, 0, nullptr))) return rc;
        FRT_HIP_CHECK(hipMemcpyAsync(y_packed, h->ypacked.ptr, out_bytes, hipMemcpyDeviceToHost, h->stream));
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        return FRT_OK;
    }
    if (dx) return h->mode == 1 ? frt_ola_filter(h, x, n, y_packed, plen) : run_stages(h, x, 0, n, n, y_packed, plen, nullptr, 0, 0);
    return filter_host(h, x, n, y_packed, plen);
}

extern "C" int frt_octbank_energies(frt_octbank* h, const float* x, int64_t n, int block, const double* alphas,
                                    const double* weight_db, int as_db, float* energy_out) {
    FRT_REQUIRE(h && h->bpo >= 1, "frt_octbank_energies: needs a handle with bands");
    // mode 1, ONE block of any length up to 1024 = the octave-spectrum widget's chunk handler (octavespectrum.py:91-122)
    const bool chunk_call = h->mode == 1 && n == block && block >= 1 && block <= 1024;
    const bool chunk_kernels = chunk_call;
    FRT_REQUIRE(chunk_call || (block >= 256 && (block & (block - 1)) == 0), "frt_octbank_energies: block %d must be a power of two >= 256", block);
    FRT_REQUIRE(h->mode == 0 || block <= 1024, "frt_octbank_energies: the FFT bank's cadence is blocks of at most 1024 samples");
    FRT_REQUIRE(n > 0 && n % block == 0 && n < (1ll << 31), "frt_octbank_energies: n must be a positive multiple of block");
    FRT_REQUIRE(h->chunk0 == 0 || h->chunk0 % block == 0 || (h->mode == 0 && block % h->chunk0 == 0),
                "frt_octbank_energies: chunk and block must divide one another");
    FRT_REQUIRE(x && alphas && energy_out, "frt_octbank_energies: null buffer");
    // A time-parallel chunk SHORTER than the block (mode 0; more, shorter recurrences: the output pass has one lane per chunk):
    // the block axis of the energy pipeline becomes the chunk — sp_b = E_b + sp_{b-1} (1 - alpha)^m holds for any partition of
    // the samples into consecutive blocks — and every sub-th smoothed value is one of the caller's.
    const int eb = (h->mode == 0 && h->chunk0 > 0 && h->chunk0 < block && !chunk_call) ? h->chunk0 : block;
    const int sub = block / eb;
    FRT_REQUIRE(sub == 1 || (n % h->chunk0 == 0 && is_device_pointer(x) && (uintptr_t)x % 16 == 0),
                "frt_octbank_energies: a chunk shorter than the block needs 16-byte aligned device input in whole chunks");
    const int nblocks = (int)(n / eb);                            // entries of the block axis
    const size_t ecount = (size_t)h->n_channels * nblocks * h->nbands;
    const size_t ocount = ecount / sub;                           // values the caller receives
    int rc;
    std::vector<double> al(alphas, alphas + h->nbands), dn(h->nbands);
    {
        long long slen[kNOctave];
        slen[0] = block;
        for (int j = 1; j < kNOctave; ++j) slen[j] = (slen[j - 1] + 1) / 2;
        for (int k = 0; k < h->nbands; ++k) {    // (1 - alpha)^m with m = the band's samples per entry (eb / dec; ceil chain for a ragged chunk)
            const int j = kNOctave - 1 - k / h->bpo;
            dn[k] = sub > 1 ? std::pow(1.0 - al[k], (double)eb / (double)(1 << j))
                            : std::pow(1.0 - al[k], (double)(chunk_call ? slen[j] : (long long)(block >> j)));
        }
    }
    if ((rc = upload_if_changed(h->alpha, h->alpha_host, al, h->stream)) || (rc = upload_if_changed(h->decay_n, h->decay_host, dn, h->stream)) ||
        (rc = h->eblock.reserve(ecount * sizeof(double))))
        return rc;
    if (weight_db) {
        std::vector<double> w(weight_db, weight_db + h->nbands);
        if ((rc = upload_if_changed(h->weight, h->weight_host, w, h->stream))) return rc;
    }
    if (!h->smooth.ptr) {
        if ((rc = h->smooth.reserve((size_t)h->n_channels * h->nbands * sizeof(double)))) return rc;
        FRT_HIP_CHECK(hipMemsetAsync(h->smooth.ptr, 0, h->smooth.bytes, h->stream));
    }
    const bool dx = is_device_pointer(x), dout = is_device_pointer(energy_out);
    FRT_REQUIRE(dx == dout, "frt_octbank_energies: input and output must both be host or both be device memory");
    const void* d_x = x;
    float* d_out = energy_out;
    // host buffers of the widget-sized calls travel through the handle's pinned blocks (from pageable memory the runtime
    // stages the copy itself and blocks the caller)
    const size_t xbytes = (size_t)h->n_channels * n * sizeof(float), obytes = ocount * sizeof(float);
    const bool pinned = !dx && xbytes <= ((size_t)1 << 20) && obytes <= ((size_t)1 << 20);
    if (!dx) {
        if ((rc = h->xin.reserve(xbytes)) || (rc = h->eout.reserve(obytes))) return rc;
        if (pinned) {
            if (xbytes > h->pin_in_bytes || obytes > h->pin_out_bytes) {
                // captured graphs of frt_octbank_filter carry the pinned addresses: they are rebuilt at their next use
                for (auto& e : h->graphs)
                    if (e.exec) (void)hipGraphExecDestroy(e.exec);
                h->graphs.clear();
            }
            if (xbytes > h->pin_in_bytes) {
                FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
                if (h->pin_in) (void)hipHostFree(h->pin_in);
                h->pin_in = nullptr;
                h->pin_in_bytes = 0;
                FRT_HIP_CHECK(hipHostMalloc(&h->pin_in, 2 * xbytes, hipHostMallocDefault));
                h->pin_in_bytes = 2 * xbytes;
            }
            if (obytes > h->pin_out_bytes) {
                FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
                if (h->pin_out) (void)hipHostFree(h->pin_out);
                h->pin_out = nullptr;
                h->pin_out_bytes = 0;
                FRT_HIP_CHECK(hipHostMalloc(&h->pin_out, 2 * obytes, hipHostMallocDefault));
                h->pin_out_bytes = 2 * obytes;
            }
            memcpy(h->pin_in, x, xbytes);
        }
        if (chunk_kernels && pinned) {
            d_x = h->pin_in;                       // the two launches read the chunk and write the band vector in place
            d_out = (float*)h->pin_out;
        } else {
            FRT_HIP_CHECK(hipMemcpyAsync(h->xin.ptr, pinned ? (const void*)h->pin_in : (const void*)x, xbytes, hipMemcpyHostToDevice, h->stream));
            d_x = h->xin.ptr;
            d_out = h->eout.as<float>();
        }
    }
    if (chunk_kernels) {
        // the widget's chunk: the running convolutions themselves, two launches (ola.hip, chunk path)
        if ((rc = frt_ola_chunk_energies(h, d_x, 1, (int)n, alphas, h->decay_n.as<double>(), h->smooth.as<double>(),
                                         weight_db ? h->weight.as<double>() : nullptr, as_db, d_out, 1)))
            return rc;
        if (!dx) {
            if (!pinned) FRT_HIP_CHECK(hipMemcpyAsync(energy_out, d_out, obytes, hipMemcpyDeviceToHost, h->stream));
            FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
            if (pinned) memcpy(energy_out, h->pin_out, obytes);
        }
        return FRT_OK;
    }
    if (h->mode == 1) rc = frt_ola_filter_batch(h, d_x, 1, n, nullptr, 0, h->eblock.as<double>(), block, nblocks, alphas);
    else rc = run_stages(h, d_x, 1, n, (int)n, nullptr, 0, h->eblock.as<double>(), eb, nblocks, sub);
    if (rc) return rc;
    if (nblocks >= 4 * kEnergySplit) {
        const int nsplit = (nblocks + kEnergySplit - 1) / kEnergySplit, threads = (h->nbands + 63) / 64 * 64;
        if ((rc = h->eseg.reserve((size_t)h->n_channels * (nsplit + 1) * h->nbands * sizeof(double)))) return rc;
        hipLaunchKernelGGL(energy_local_kernel, dim3(nsplit, h->n_channels), dim3(threads), 0, h->stream, h->eblock.as<double>(),
                           h->decay_n.as<double>(), h->eseg.as<double>(), h->smooth.as<double>(), nblocks, h->nbands);
        auto finish = as_db ? energy_finish_kernel<true, true> : energy_finish_kernel<false, true>;
        hipLaunchKernelGGL(finish, dim3(nsplit, h->n_channels), dim3(threads), 0, h->stream, h->eblock.as<double>(),
                           h->decay_n.as<double>(), h->eseg.as<double>(), h->smooth.as<double>(), (void*)d_out, nblocks, h->nbands,
                           weight_db ? h->weight.as<double>() : nullptr, sub);
    } else {
        hipLaunchKernelGGL(energy_scan_kernel, dim3(h->n_channels), dim3(kEnergyThreads), 0, h->stream, h->eblock.as<double>(),
                           h->decay_n.as<double>(), h->smooth.as<double>(), (void*)d_out, 1, nblocks, h->nbands,
                           weight_db ? h->weight.as<double>() : nullptr, as_db, sub);
    }
    FRT_HIP_CHECK(hipGetLastError());
    if (!dx) {
        FRT_HIP_CHECK(hipMemcpyAsync(pinned ? h->pin_out : (void*)energy_out, d_out, obytes, hipMemcpyDeviceToHost, h->stream));
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (pinned) memcpy(energy_out, h->pin_out, obytes);
    }
    return FRT_OK;
}

// ---- G2: stand-alone decimation chain (decimate_multiple, friture/signal/decimate.py:45-71) -------
// A handle with bands_per_octave = 0 carries only the decimator; `n_stages` of its nine stages run.
extern "C" int frt_decimate_multiple(frt_octbank* h, int n_stages, const double* x, int n, double* out, int* n_out) {
    FRT_REQUIRE(h && h->bpo == 0, "frt_decimate_multiple: needs a handle created with bands_per_octave = 0");
    FRT_REQUIRE(n_stages >= 1 && n_stages < kNOctave, "frt_decimate_multiple: n_stages %d not in [1, 8]", n_stages);
    FRT_REQUIRE(n >= 0, "frt_decimate_multiple: n < 0");
    int len[kNOctave];
    stage_lengths(n, len);
    if (n_out) *n_out = len[n_stages];
    if (n == 0) return FRT_OK;                  // decimate.py:56-57: empty input is passed through
    FRT_REQUIRE(x && out, "frt_decimate_multiple: null buffer");
    const bool dx = is_device_pointer(x), dout = is_device_pointer(out);
    FRT_REQUIRE(dx == dout, "frt_decimate_multiple: input and output must both be host or both be device memory");
    int rc;
    const void* d_x = x;
    if (!dx) {
        if ((rc = h->xin.reserve((size_t)h->n_channels * n * sizeof(double)))) return rc;
        FRT_HIP_CHECK(hipMemcpyAsync(h->xin.ptr, x, (size_t)h->n_channels * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_x = h->xin.ptr;
    }
    for (int j = 1; j <= n_stages; ++j)
        if ((rc = h->xbuf[j].reserve((size_t)h->n_channels * len[j] * sizeof(double)))) return rc;
    for (int j = 0; j < n_stages; ++j) {
        IirStageArgs a{};
        a.x = j == 0 ? d_x : h->xbuf[j].ptr;
        a.x_stride = len[j];
        a.n = len[j];
        a.coef = h->coef.as<double>();
        a.order = h->order.as<int>();
        a.nfilt = 1;
        a.dec_filter = 0;
        a.state = h->state.as<double>() + (size_t)j * h->stage_state_elems();
        a.chunk = (len[j] + 63) / 64 * 64;
        a.nchunks = 1;
        a.pass = 0;
        a.band_index[0] = -1;
        a.xnext = h->xbuf[j + 1].as<double>();
        a.xnext_stride = len[j + 1];
        if ((rc = launch_iir_stage(a, h->h_order.data(), h->n_channels, h->stream))) return rc;
    }
    const size_t obytes = (size_t)h->n_channels * len[n_stages] * sizeof(double);
    FRT_HIP_CHECK(hipMemcpyAsync(out, h->xbuf[n_stages].ptr, obytes, dx ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    if (!dx) FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    return FRT_OK;
}

// decimate_multiple with the reference's functional interface (decimate.py:45-71: the filter states are arguments and
// results) as ONE call on host arrays: samples and states are read, and the decimated signal and the new states written,
// in place in page-locked memory by the stage kernels — a call that went through frt_octbank_set_state,
// frt_decimate_multiple and frt_octbank_get_state made three round trips.  One channel; zi / zf: [n_stages][12] or NULL
// (zero state / states not wanted).  The handle's own carried state is not touched.
extern "C" int frt_decimate_multiple_state(frt_octbank* h, int n_stages, const double* x, int n, const double* zi, double* out, int* n_out,
                                           double* zf) {
    FRT_REQUIRE(h && h->bpo == 0, "frt_decimate_multiple_state: needs a handle created with bands_per_octave = 0");
    FRT_REQUIRE(n_stages >= 1 && n_stages < kNOctave, "frt_decimate_multiple_state: n_stages %d not in [1, 8]", n_stages);
    FRT_REQUIRE(n >= 0, "frt_decimate_multiple_state: n < 0");
    int len[kNOctave];
    stage_lengths(n, len);
    if (n_out) *n_out = len[n_stages];
    if (n == 0) return FRT_OK;
    FRT_REQUIRE(x && out && !is_device_pointer(x) && !is_device_pointer(out), "frt_decimate_multiple_state: host arrays");
    // every channel of the handle in the same launches (the delay estimator decimates its two channels chunk by chunk:
    // delay_estimator.py:97-98 — two calls of two dependent launches each were two device round trips per chunk)
    const int C = h->n_channels;
    const int ord = h->h_order[0];
    const size_t xbytes = (size_t)C * n * sizeof(double), sbytes = (size_t)n_stages * C * kStates * sizeof(double);
    const size_t obytes = (size_t)C * len[n_stages] * sizeof(double);
    const size_t in_need = (xbytes + 255) / 256 * 256 + sbytes;
    FRT_REQUIRE(in_need <= kZeroCopyMax && obytes <= kZeroCopyMax, "frt_decimate_multiple_state: %d samples x %d channels exceed the in-place call", n, C);
    int rc;
    if (in_need > h->pin_in_bytes || obytes > h->pin_out_bytes) {
        FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
        for (auto& e : h->graphs)                              // captured graphs carry the pinned addresses
            if (e.exec) (void)hipGraphExecDestroy(e.exec);
        h->graphs.clear();
        if (in_need > h->pin_in_bytes) {
            if (h->pin_in) (void)hipHostFree(h->pin_in);
            h->pin_in = nullptr;
            h->pin_in_bytes = 0;
            FRT_HIP_CHECK(hipHostMalloc(&h->pin_in, 2 * in_need, hipHostMallocDefault));
            h->pin_in_bytes = 2 * in_need;
        }
        if (obytes > h->pin_out_bytes) {
            if (h->pin_out) (void)hipHostFree(h->pin_out);
            h->pin_out = nullptr;
            h->pin_out_bytes = 0;
            FRT_HIP_CHECK(hipHostMalloc(&h->pin_out, 2 * obytes, hipHostMallocDefault));
            h->pin_out_bytes = 2 * obytes;
        }
    }
    double* px = (double*)h->pin_in;
    double* ps = (double*)((char*)h->pin_in + (xbytes + 255) / 256 * 256);      // [stage][channel][kStates]: the kernel's layout
    memcpy(px, x, xbytes);
    for (int j = 0; j < n_stages; ++j)
        for (int c = 0; c < C; ++c)
            for (int s = 0; s < kStates; ++s)
                ps[((size_t)j * C + c) * kStates + s] = (zi && s < ord) ? zi[((size_t)c * n_stages + j) * ord + s] : 0.0;
    for (int j = 1; j < n_stages; ++j)
        if ((rc = h->xbuf[j].reserve((size_t)C * len[j] * sizeof(double)))) return rc;
    for (int j = 0; j < n_stages; ++j) {
        IirStageArgs a{};
        a.x = j == 0 ? (const void*)px : h->xbuf[j].ptr;
        a.x_stride = len[j];
        a.n = len[j];
        a.coef = h->coef.as<double>();
        a.order = h->order.as<int>();
        a.nfilt = 1;
        a.dec_filter = 0;
        a.state = ps + (size_t)j * C * kStates;
        a.chunk = (len[j] + 63) / 64 * 64;
        a.nchunks = 1;
        a.pass = 0;
        a.band_index[0] = -1;
        a.xnext = j + 1 == n_stages ? (double*)h->pin_out : h->xbuf[j + 1].as<double>();
        a.xnext_stride = len[j + 1];
        if ((rc = launch_iir_stage(a, h->h_order.data(), C, h->stream))) return rc;
    }
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    memcpy(out, h->pin_out, obytes);
    if (zf)
        for (int j = 0; j < n_stages; ++j)
            for (int c = 0; c < C; ++c)
                for (int s = 0; s < ord; ++s) zf[((size_t)c * n_stages + j) * ord + s] = ps[((size_t)j * C + c) * kStates + s];
    return FRT_OK;
}

// ---- lfilter_float64_1D (friture/signal/lfilter.py:85-147): one filter, explicit state in / out ----
extern "C" int frt_lfilter_f64(const double* b, const double* a, int n_coef, const double* x, int n, const double* zi,
                               double* y, double* zf) {
    FRT_REQUIRE(b && a && n_coef >= 1 && n_coef <= kMaxOrder + 1, "frt_lfilter_f64: 1 <= len(b) = len(a) <= %d required",
                kMaxOrder + 1);
    FRT_REQUIRE(n >= 0 && (n == 0 || (x && y)), "frt_lfilter_f64: bad buffers");
    FRT_REQUIRE(n_coef == 1 || (zi && zf), "frt_lfilter_f64: state vectors required");
    const int order = n_coef - 1;
    if (order == 0) {                                   // lfilter.py:140-142: pure gain
        for (int k = 0; k < n; ++k) y[k] = x[k] * b[0];
        return FRT_OK;
    }
    if (n == 0) {
        for (int s = 0; s < order; ++s) zf[s] = zi[s];
        return FRT_OK;
    }
    std::vector<double> coef(kCoefStride, 0.0), st(kStates, 0.0);
    for (int t = 0; t < n_coef; ++t) {
        coef[t] = b[t];
        coef[kMaxOrder + 1 + t] = a[t];
    }
    for (int s = 0; s < order; ++s) st[s] = zi[s];
    // Arguments travel through the process-level staging arena (StageCall, common.h): one pinned block, one asynchronous
    // upload and download on its stream — or none at all for widget-sized calls, which the kernel serves in place from
    // page-locked memory.  (Rounds 2-3 kept five grow-only buffers per calling THREAD: their destructor ran hipFree at
    // thread / process exit, possibly after the runtime's own teardown, and the copies were blocking null-stream ones.)
    std::vector<int> ord(1, order);
    StageCall sc;
    const int icoef = sc.add_in(coef.data(), coef.size() * sizeof(double));
    const int iord = sc.add_in(ord.data(), sizeof(int));
    const int ist0 = sc.add_in(st.data(), kStates * sizeof(double));
    const int ix = sc.add_in(x, (size_t)n * sizeof(double));
    const int iy = sc.add_out(y, (size_t)n * sizeof(double));
    const int ist = sc.add_out(st.data(), kStates * sizeof(double));      // the kernel updates its state block in place
    int rc = sc.begin();
    if (rc) return rc;
    FRT_HIP_CHECK(hipMemcpyAsync(sc.ptr<double>(ist), sc.ptr<double>(ist0), kStates * sizeof(double), hipMemcpyDefault, sc.stream()));
    IirStageArgs s{};
    s.x = sc.ptr<double>(ix);
    s.x_stride = n;
    s.n = n;
    s.coef = sc.ptr<double>(icoef);
    s.order = sc.ptr<int>(iord);
    s.nfilt = 1;
    s.dec_filter = -1;
    s.state = sc.ptr<double>(ist);
    s.chunk = (n + 63) / 64 * 64;
    s.nchunks = 1;
    s.pass = 0;
    s.y = sc.ptr<double>(iy);
    s.y_cstride = n;
    s.band_index[0] = -1;
    if ((rc = launch_iir_stage(s, &order, 1, sc.stream()))) return rc;
    if ((rc = sc.finish())) return rc;
    for (int t = 0; t < order; ++t) zf[t] = st[t];
    return FRT_OK;
}

// ---- the delay estimator's per-chunk part as one device-resident object --------------------------------------------
// Delay_Estimator_Widget.handle_new_data (friture/delay_estimator.py:87-131), per chunk of both channels: two chained
// decimations by 2 with carried state (decimate_multiple, :97-98), push into the two private ring buffers (:99-100), and —
// once per `needed` decimated samples, 94 chunks at the default range — a window of both rings for GCC-PHAT.  Round 2's
// stream class drove this through torch (an upload from pageable memory, two allocations, the decimation launches, two
// ring pushes of several torch kernels each: 124 us per chunk against 30 us of numpy).  Here the chunk goes through a
// pinned slot that the first decimation stage reads in place (zero copy), both stages and ONE ring-write launch are
// enqueued on the object's stream, and the call returns without waiting: nothing comes back per chunk.  Windows are handed
// out as device pointers into the mirror rings (ringbuffer.py:87-99: the `length` samples ending at an absolute index), with
// the two helpers the reference's window handling needs: the standard deviations of the gate (:127) and the in-place mean
// removal generalized_cross_correlation applies to its views (correlation.py:27-28).
namespace frt {

__global__ void __launch_bounds__(256) delay_ring_write_kernel(const double* __restrict__ dec, int m, double* __restrict__ ring,
                                                               long long ring_len, long long offset) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (t >= m) return;
    const long long p = (offset + t) % ring_len;
    const double v = dec[(size_t)c * m + t];
    double* r = ring + (size_t)c * 2 * ring_len;
    r[p] = v;                               // ringbuffer.py:52-59: the second copy makes every window a linear view
    r[p + ring_len] = v;
}

// RingBuffer.grow_if_needed (ringbuffer.py:102-130) moves the old ring PHYSICALLY: its first copy lands `shift` positions
// further (shift chosen so that the write head keeps its absolute index), the mirror copy follows and folds around the
// end.  Samples that had wrapped in the old ring therefore do not land where their absolute index would put them; the
// windows the reference hands out afterwards contain exactly that, so the same three slice copies are made here.
__global__ void __launch_bounds__(256) delay_ring_relay_kernel(const double* __restrict__ old_ring, long long old_len,
                                                               double* __restrict__ new_ring, long long new_len, long long shift) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (p >= old_len) return;
    const double v = old_ring[(size_t)c * 2 * old_len + p];
    double* r = new_ring + (size_t)c * 2 * new_len;
    const long long direct = old_len < new_len - shift ? old_len : new_len - shift;
    r[shift + p] = v;                                   // :120 first copy, always complete
    if (p < direct) r[new_len + shift + p] = v;         // :124 second copy ...
    else r[p - direct] = v;                             // :125 ... folded
}

// numpy.std of a window (population, two passes: mean, then mean of squared deviations), one workgroup per channel
__global__ void __launch_bounds__(1024) delay_window_std_kernel(const double* __restrict__ d0, const double* __restrict__ d1, int length,
                                                                double* __restrict__ out) {
    __shared__ double red[16];
    const double* d = blockIdx.x == 0 ? d0 : d1;
    const int tid = threadIdx.x;
    auto block_sum = [&](double v) -> double {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w];
        return s;
    };
    // a constant window is silent: numpy.std of n copies of c is 0 or rounding noise of c, depending on c, n and the
    // summation order (delay_estimator.py:127 gates on std > 0) — here "every sample equals the first" decides
    const double first = d[0];
    double acc = 0.0, differ = 0.0;
    for (int i = tid; i < length; i += 1024) {
        const double v = d[i];
        acc += v;
        differ += v != first ? 1.0 : 0.0;
    }
    const double mean = block_sum(acc) / (double)length;
    const bool constant = block_sum(differ) == 0.0;
    acc = 0.0;
    for (int i = tid; i < length; i += 1024) {
        const double t = d[i] - mean;
        acc += t * t;
    }
    const double var = constant ? 0.0 : block_sum(acc) / (double)length;
    if (tid == 0) out[blockIdx.x] = sqrt(var);
}

// d -= mean on a window of a mirror ring: the window's samples and their mirror images (the reference's view aliases the
// ring storage, whose two halves its push keeps identical only for the samples it writes — the mirror copy of a de-meaned
// sample is NOT updated there either: ringbuffer.py:87-99 returns buffer[:, start:stop] of the doubled array)
__global__ void __launch_bounds__(256) delay_demean_kernel(double* __restrict__ w0, double* __restrict__ w1, int length,
                                                           const double* __restrict__ means) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= length) return;
    if (blockIdx.y == 0) w0[i] -= means[0];
    else w1[i] -= means[1];
}

}  // namespace frt

struct frt_delay {
    frt_octbank* dec = nullptr;             // two channels, decimator only
    int n_stages = 2;
    hipStream_t stream = nullptr;
    DeviceBuffer ring;                      // [2][2 ring_len]
    long long ring_len = 0, offset = 0;     // offset: decimated samples pushed so far
    static constexpr int kSlots = 4;        // pinned chunks in flight
    char* pin[kSlots] = {};
    size_t pin_bytes[kSlots] = {};
    hipEvent_t done[kSlots] = {};
    bool pending[kSlots] = {};
    int slot = 0;
    DeviceBuffer stats, xin;
};

extern "C" void frt_delay_destroy(frt_delay* h) {
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->dec) frt_octbank_destroy(h->dec);
    h->ring.release();
    h->stats.release();
    h->xin.release();
    for (int s = 0; s < frt_delay::kSlots; ++s) {
        if (h->pin[s]) (void)hipHostFree(h->pin[s]);
        if (h->done[s]) (void)hipEventDestroy(h->done[s]);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int frt_delay_create(frt_delay** out, const double* bdec, const double* adec, int n_stages, int ring_length) {
    FRT_REQUIRE(out && bdec && adec, "frt_delay_create: null argument");
    *out = nullptr;
    FRT_REQUIRE(n_stages >= 1 && n_stages < kNOctave && ring_length >= 16, "frt_delay_create: bad n_stages %d / ring length %d", n_stages,
                ring_length);
    frt_delay* h = new frt_delay();
    h->n_stages = n_stages;
    h->ring_len = ring_length;
    int rc = frt_octbank_create(&h->dec, 0, 2, 0, nullptr, nullptr, bdec, adec, nullptr, nullptr);
    if (!rc && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = FRT_ERR_HIP;
    if (!rc) rc = frt_octbank_set_stream(h->dec, h->stream);
    for (int s = 0; !rc && s < frt_delay::kSlots; ++s)
        if (hipEventCreateWithFlags(&h->done[s], hipEventDisableTiming) != hipSuccess) rc = FRT_ERR_HIP;
    if (!rc) rc = h->ring.reserve((size_t)2 * 2 * ring_length * sizeof(double));
    if (!rc) rc = h->stats.reserve(2 * sizeof(double));
    if (!rc && hipMemsetAsync(h->ring.ptr, 0, h->ring.bytes, h->stream) != hipSuccess) rc = FRT_ERR_HIP;
    if (rc) {
        frt_delay_destroy(h);
        return rc;
    }
    *out = h;
    return FRT_OK;
}

extern "C" void* frt_delay_stream(frt_delay* h) { return h ? (void*)h->stream : nullptr; }

// The rings hold at least `length` samples from now on (RingBuffer.grow_if_needed, ringbuffer.py:102-130: x1.5).
extern "C" int frt_delay_reserve(frt_delay* h, int length) {
    FRT_REQUIRE(h && length >= 1, "frt_delay_reserve: bad arguments");
    if (length <= h->ring_len) return FRT_OK;
    const long long new_len = (long long)(1.5 * length);
    DeviceBuffer grown;
    int rc;
    if ((rc = grown.reserve((size_t)2 * 2 * new_len * sizeof(double)))) return rc;
    FRT_HIP_CHECK(hipMemsetAsync(grown.ptr, 0, grown.bytes, h->stream));
    const long long shift = ((h->offset % new_len - h->offset % h->ring_len) % new_len + new_len) % new_len;
    hipLaunchKernelGGL(delay_ring_relay_kernel, dim3((unsigned)((h->ring_len + 255) / 256), 2), dim3(256), 0, h->stream,
                       h->ring.as<double>(), h->ring_len, grown.as<double>(), new_len, shift);
    FRT_HIP_CHECK(hipGetLastError());
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->ring.release();
    h->ring = grown;
    grown.ptr = nullptr;
    grown.bytes = 0;
    h->ring_len = new_len;
    return FRT_OK;
}

// One chunk of both channels: x [2][n] float64 in host memory.  Returns (offset_out) the number of decimated samples in
// the rings afterwards.  Asynchronous: the work is enqueued on the object's stream, nothing is waited for.
extern "C" int frt_delay_push(frt_delay* h, const double* x, int n, int64_t* offset_out) {
    FRT_REQUIRE(h && n >= 0, "frt_delay_push: bad arguments");
    if (offset_out) *offset_out = h->offset;
    if (n == 0) return FRT_OK;
    FRT_REQUIRE(x && !is_device_pointer(x), "frt_delay_push: x must be a host array");
    int len[kNOctave];
    stage_lengths(n, len);
    const int m = len[h->n_stages];
    int rc;
    if (m > h->ring_len && (rc = frt_delay_reserve(h, m))) return rc;
    // the chunk travels through a pinned slot (an asynchronous copy needs page-locked memory to be asynchronous)
    const int s = h->slot;
    h->slot = (s + 1) % frt_delay::kSlots;
    if (h->pending[s]) {
        FRT_HIP_CHECK(hipEventSynchronize(h->done[s]));
        h->pending[s] = false;
    }
    const size_t xbytes = (size_t)2 * n * sizeof(double);
    if (xbytes > h->pin_bytes[s]) {
        if (h->pin[s]) (void)hipHostFree(h->pin[s]);
        h->pin[s] = nullptr;
        h->pin_bytes[s] = 0;
        FRT_HIP_CHECK(hipHostMalloc((void**)&h->pin[s], 2 * xbytes, hipHostMallocDefault));
        h->pin_bytes[s] = 2 * xbytes;
    }
    memcpy(h->pin[s], x, xbytes);
    if ((rc = h->xin.reserve(xbytes))) return rc;
    FRT_HIP_CHECK(hipMemcpyAsync(h->xin.ptr, h->pin[s], xbytes, hipMemcpyHostToDevice, h->stream));
    FRT_HIP_CHECK(hipEventRecord(h->done[s], h->stream));
    h->pending[s] = true;
    frt_octbank* d = h->dec;
    for (int j = 1; j <= h->n_stages; ++j)
        if ((rc = d->xbuf[j].reserve((size_t)2 * len[j] * sizeof(double)))) return rc;
    for (int j = 0; j < h->n_stages; ++j) {
        IirStageArgs a{};
        a.x = j == 0 ? h->xin.ptr : d->xbuf[j].ptr;
        a.x_stride = len[j];
        a.n = len[j];
        a.coef = d->coef.as<double>();
        a.order = d->order.as<int>();
        a.nfilt = 1;
        a.dec_filter = 0;
        a.state = d->state.as<double>() + (size_t)j * d->stage_state_elems();
        a.chunk = (len[j] + 63) / 64 * 64;
        a.nchunks = 1;
        a.pass = 0;
        a.band_index[0] = -1;
        a.xnext = d->xbuf[j + 1].as<double>();
        a.xnext_stride = len[j + 1];
        if ((rc = launch_iir_stage(a, d->h_order.data(), 2, h->stream))) return rc;
    }
    hipLaunchKernelGGL(delay_ring_write_kernel, dim3((m + 255) / 256, 2), dim3(256), 0, h->stream, d->xbuf[h->n_stages].as<double>(), m,
                       h->ring.as<double>(), h->ring_len, h->offset);
    FRT_HIP_CHECK(hipGetLastError());
    h->offset += m;
    if (offset_out) *offset_out = h->offset;
    return FRT_OK;
}

// Device pointers of the two windows of `length` samples ending at absolute index `end` (RingBuffer.data_indexed).
extern "C" int frt_delay_window(frt_delay* h, int64_t end, int length, double** d0, double** d1) {
    FRT_REQUIRE(h && d0 && d1 && length >= 1, "frt_delay_window: bad arguments");
    FRT_REQUIRE(end >= 0 && end <= h->offset, "frt_delay_window: end index %lld outside [0, %lld]", (long long)end, (long long)h->offset);
    int rc;
    const long long need = length + h->offset - end;              // ringbuffer.py:96: grow_if_needed(length + offset - start)
    FRT_REQUIRE(need < (1ll << 30), "frt_delay_window: window too far back");
    if (need > h->ring_len && (rc = frt_delay_reserve(h, (int)need))) return rc;
    const long long stop = end % h->ring_len + h->ring_len;       // ringbuffer.py:91-92
    *d0 = h->ring.as<double>() + (stop - length);
    *d1 = h->ring.as<double>() + 2 * h->ring_len + (stop - length);
    return FRT_OK;
}

// numpy.std of both windows (the gate of delay_estimator.py:127); waits for everything enqueued before.
extern "C" int frt_delay_window_std(frt_delay* h, const double* d0, const double* d1, int length, double* std_out) {
    FRT_REQUIRE(h && d0 && d1 && std_out && length >= 1, "frt_delay_window_std: bad arguments");
    hipLaunchKernelGGL(delay_window_std_kernel, dim3(2), dim3(1024), 0, h->stream, d0, d1, length, h->stats.as<double>());
    FRT_HIP_CHECK(hipGetLastError());
    FRT_HIP_CHECK(hipMemcpyAsync(std_out, h->stats.ptr, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    FRT_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int s = 0; s < frt_delay::kSlots; ++s) h->pending[s] = false;
    return FRT_OK;
}

// The in-place mean removal of generalized_cross_correlation (correlation.py:27-28) on the ring views.
// `means`: two doubles in device memory (what frt_gcc_phat wrote); `stream`: the stream frt_gcc_phat ran on (the removal
// must follow its reads of the windows), null = the object's own.
extern "C" int frt_delay_demean(frt_delay* h, double* d0, double* d1, int length, const double* means, void* stream) {
    FRT_REQUIRE(h && d0 && d1 && means && length >= 1, "frt_delay_demean: bad arguments");
    hipLaunchKernelGGL(delay_demean_kernel, dim3((length + 255) / 256, 2), dim3(256), 0, stream ? (hipStream_t)stream : h->stream, d0, d1,
                       length, means);
    FRT_HIP_CHECK(hipGetLastError());
    return FRT_OK;
}
