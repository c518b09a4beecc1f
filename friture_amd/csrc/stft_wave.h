// stft_wave.h — device side of K1 for N <= 1024 (one frame per lane group inside a wavefront): StftArgs, the output
// epilogues shared with the large-frame instances, and stft_kernel.  Split out of stft.hip so that the digest bench.py
// stamps PMC traffic figures with (bench.kernel_source_digest: this file + fft_core.h) changes when the measured kernel
// changes, not when the host-side launcher next to it does.  Included by stft.hip only.
#pragma once
#include <cmath>
#include <type_traits>

#include "common.h"
#include "fft_core.h"

namespace frt {

struct StftArgs {
    const void* x;         // [C][x_stride] samples
    void* out;             // [C][F][M+1]; split layout: [C][F][M], bins 0..M-1 (every row a whole number of 64-byte lines)
    void* out_nyq;         // split layout: [C][F], bin M of every frame; null = packed rows
    const void* window;    // [N] T
    const void* tw;        // [M] cpx<T>: exp(-2 pi i n / M)
    const void* twn;       // [M] cpx<T>: exp(-2 pi i k / N)
    const void* tws;       // [M/16] cpx<T>: exp(-2 pi i n / (M/16)), sub-transforms of stft_big_kernel (N >= 2048)
    const void* weight;    // [M+1] T or null
    const void* wimage;    // [M+1] T: 255 (weight - spec_min)/(spec_max - spec_min), colour-index offset per bin
    double image_gain;     // 255 * (10/log2(10)) / (spec_max - spec_min): colour index per log2 of the PSD
    const uint32_t* lut;   // [256] or null
    long long x_stride;    // elements between channels
    long long n_frames;    // frames per channel
    long long frame_base;  // first frame this launch handles
    long long out_cstride; // elements between channels in out
    int hop;
    int run;               // frames per run
    int runs_per_channel;
    int n_groups;          // total lane groups = C * runs_per_channel
    int kind;              // FRT_STFT_*
    int vec2;              // 1: 2-sample vector loads are aligned
    double norm_off;       // -spec_min
    double norm_scale;     // 1 / (spec_max - spec_min)
    // float32 IMAGE kind, exact colour indices (see exact_colour_index)
    const double* edge_pow;// [256] power at which the unweighted index value reaches n: 10^((min + n (max - min)/255)/10)
    const double* bin_pow; // [M+1] 10^(-weight[k]/10)
    float edge2;           // width of the zone above an index edge that is decided in float64
    double image_thr;      // float64 instance: the margin the float32 table of the float32 instance carries inside wimage
    int rising;            // max > min: the index grows with the power
    int eps_free;          // P + 1e-30 == P in float32 wherever it matters: the add is skipped
#ifdef FRT_ABLATE
    int ablate;            // experiment switches: 1 no stores, 2 no loads, 4 no FFT, 8 no unpack shuffles
#endif
};


template <typename T> __device__ __forceinline__ T db10(T p);
template <> __device__ __forceinline__ float db10<float>(float p) {
    // 10*log10(v) = (10/log2(10)) * log2(v); v >= 1e-30 is a normal float, v_log_f32 is exact enough
    return 3.01029995663981195f * __log2f(p + 1e-30f);
}
template <> __device__ __forceinline__ double db10<double>(double p) { return 10.0 * log10(p + 1e-30); }
__device__ __forceinline__ float log2_t(float v) { return __log2f(v); }
__device__ __forceinline__ double log2_t(double v) { return log2(v); }

template <typename T>
__device__ __forceinline__ T shfl_t(T v, int lane) { return __shfl(v, lane, 64); }

// ---- colour index of the IMAGE kind --------------------------------------------------------------------------
// The reference computes, in float64 (spectrogram.py:119-129, color_tranform.py:48-51, lookup_table.py:50-52),
//     idx = int(clip((10 log10(P + 1e-30) + w[k] - min) / (max - min), 0, 1) * 255).
// The float32 instances evaluate q = gain * log2(P + 1e-30) + wimage[k] (one v_log_f32 and one fma per bin;
// wimage carries weight, range and a margin `thr`), whose distance from the float64 value is below `thr` (bound
// derived in frt_stft_set_epilogue).  So floor(q) IS the reference's index unless q lies within 2 thr above an
// integer n (about 3e-4 of all bins); there the index is n or n - 1, and because the reference's expression is
// monotonic in P the choice is one float64 comparison:  idx >= n  <=>  P + 1e-30 >= 10^((min + n (max-min)/255 - w[k])/10)
// = edge_pow[n] * bin_pow[k] (host tables, 1e-16 relative).  The colour index is therefore exact given the float32
// power P — what remains against the reference's image is the float32 transform's own error in P.
// Such bins are served one at a time with wave-uniform operands: the two table values arrive through SCALAR loads
// (lgkmcnt).  A vector load here would have to be waited for with vmcnt(0), i.e. behind the acknowledgement of every row
// store still in flight — measured: +7 % on the whole kernel for a path that one frame in five enters.
__device__ __forceinline__ double readlane_power(float p, int src) {
    return (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p), src));
}
__device__ __forceinline__ double readlane_power(double p, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(p), src), hi = __builtin_amdgcn_readlane(__double2hiint(p), src);
    return __hiloint2double(hi, lo);
}
template <typename TP>
__device__ __forceinline__ int exact_colour_index(bool near_edge, TP p, int k, int n, const StftArgs& a) {
    unsigned long long todo = __ballot(near_edge);
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int ks = __builtin_amdgcn_readlane(k, src), ns = __builtin_amdgcn_readlane(n, src);
        const double ps = readlane_power(p, src);
        // constant address space + uniform index = s_load_dwordx2 (the tables are never written by a kernel)
        typedef const double __attribute__((address_space(4))) * ktable;
        const double edge = ((ktable)(uintptr_t)a.edge_pow)[ns] * ((ktable)(uintptr_t)a.bin_pow)[ks];
        const bool at_least = (ps + 1e-30 >= edge) == (a.rising != 0);
        if (lane == src) n = at_least ? ns : ns - 1;
    }
    return n;
}
// q clamped into the LUT's range: [0, 0.5) and everything below share index 0 (no edge there), 255.5 maps to 255
__device__ __forceinline__ float clamp_index(float q) { return __builtin_amdgcn_fmed3f(q, 0.5f, 255.5f); }
__device__ __forceinline__ double clamp_index(double q) { return fmin(fmax(q, 0.5), 255.5); }

// Row stores.  Plain by default.  Non-temporal stores (-DFRT_NT_STORES) keep the rows from evicting the samples
// out of L2 / the Infinity Cache: a batch that is re-used or was just produced is then read 15 % faster, but a batch
// read cold from HBM — the benchmark's case — runs 3 % slower and writes 5 % more bytes (the L2 lets go of
// partially written lines earlier).  Measured at sustained clocks, DESIGN.md §5/§6.
template <typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
#ifdef FRT_NT_STORES
    __builtin_nontemporal_store(v, p);
#elif defined(FRT_STORE_POLICY)          // cache-policy experiments: the bits as assembler text, e.g. -DFRT_STORE_POLICY='"sc1"'
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
    if constexpr (sizeof(T) == 4) asm volatile("global_store_dword %0, %1, off " FRT_STORE_POLICY :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off " FRT_STORE_POLICY :: "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}

// acc[lane] = the first active lane's v, other lanes of acc untouched (`lane` wave-uniform, `me` this lane's index)
__device__ __forceinline__ uint32_t lane_insert(uint32_t acc, uint32_t v, int lane, int me) {
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    return me == lane ? b : acc;
}
__device__ __forceinline__ uint32_t value_bits_lo(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint32_t value_bits_lo(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t value_bits_lo(double v) { return (uint32_t)__double2loint(v); }
__device__ __forceinline__ uint32_t value_bits_hi(float) { return 0u; }
__device__ __forceinline__ uint32_t value_bits_hi(uint32_t) { return 0u; }
__device__ __forceinline__ uint32_t value_bits_hi(double v) { return (uint32_t)__double2hiint(v); }

// TIN: sample type in HBM; T: arithmetic type; SHIFT: register slots a hop advances (0 = reload all)
// SPLIT: the split output layout (frt_stft_run_split): rows of M values, bins 0..M-1, so that every row is a whole number of
// 64-byte lines and every wave-wide store of one-wavefront frames covers four whole lines (the packed layout's rows of M + 1
// values start on 4-byte boundaries: a 256-byte store touches five lines, TCP_TCC_WRITE_REQ x1.25 of the lines written);
// bin M (the Nyquist bin) of every frame goes to a plane of its own, collected in a register across the run
// three waves per SIMD for the one-wave-per-frame instances: the float64 fix-up of the IMAGE kind would otherwise push the
// N = 512 / 1024 instances two registers over the 168 that three waves allow (no spills at hop N/2 and N/4)
#ifndef FRT_WAVE_MIN_WAVES
#define FRT_WAVE_MIN_WAVES 3
#endif
// float64: every hoisted constant and every point is a register pair; three waves per SIMD (168 registers) spilled 133 of
// them to scratch (448 bytes per lane: 0.34 ms for 65 535 frames of N = 1024, a fifth of the HBM roofline)
#ifndef FRT_WAVE_MIN_WAVES_F64
#define FRT_WAVE_MIN_WAVES_F64 2
#endif
// SHIFT = -1: the RING instance (float32, hop = N/2, one wavefront per frame, 16-byte aligned rows).  The frame's samples
// do not live in a register window: each wavefront owns a ring of two half-frames in LDS, the half-frame that the frame
// after next needs is copied there from HBM by an LDS-DMA issued from inline assembly (stft_big.h explains why) as soon as
// the window multiply has read the half it replaces, and the window multiply reads its eight slots with ds_read_b64.
// Measured at three waves per SIMD (139 VGPRs): PSD kind +1.5 ... +4 % depending on the session, colour kind equal to 0.7 %
// behind (its VALU pipe is 97 % busy either way; it keeps the register window).  At four waves (128 VGPRs, 20-36 bytes of
// spills, whose scratch traffic also counts against the hand-placed vmcnt) 4 % slower than the register-window instance.
#ifndef FRT_RING_MIN_WAVES
#define FRT_RING_MIN_WAVES 3
#endif
#ifndef FRT_RING_HALVES          // half-frames in a wavefront's ring: 2 (the copy has one frame to land) or 3 (two frames)
#define FRT_RING_HALVES 2
#endif
#ifndef FRT_RING_WEIGHTS_IN_LDS
#define FRT_RING_WEIGHTS_IN_LDS 0
#endif
template <typename TIN, typename T, int LOG2M, int SHIFT, bool SPLIT = false>
__global__ void
#if defined(FRT_WAVE_MIN_WAVES)
__launch_bounds__((Pow2Plan<LOG2M>::TPF < 256 ? 256 : Pow2Plan<LOG2M>::TPF),
                  (Pow2Plan<LOG2M>::TPF <= 64 ? (SHIFT < 0 ? FRT_RING_MIN_WAVES : sizeof(T) == 8 ? FRT_WAVE_MIN_WAVES_F64 : FRT_WAVE_MIN_WAVES) : 1))
#else
__launch_bounds__((Pow2Plan<LOG2M>::TPF < 256 ? 256 : Pow2Plan<LOG2M>::TPF))
#endif
stft_kernel(const StftArgs a) {
    using P = Pow2Plan<LOG2M>;
    constexpr int M = P::M, TPF = P::TPF;            // M = N/2 complex points
    constexpr int BLOCK = TPF < 256 ? 256 : TPF;
    constexpr int GPB = BLOCK / TPF;                 // lane groups (concurrent frames) per block
    constexpr bool WAVE = TPF <= 64;                 // a frame lives inside one wavefront
    using C = cpx<T>;
    using CIN = cpx<TIN>;

    constexpr bool RING = SHIFT < 0;
    static_assert(!RING || (TPF == 64 && sizeof(T) == 4 && sizeof(TIN) == 4), "ring instance: one wavefront per float32 frame");
    __shared__ C lds[GPB * lds_padded_size(M)];
    __shared__ uint32_t lut_lds[256];                // colour words: gathered per bin, keep them on-chip
    constexpr int NH = FRT_RING_HALVES;
    __shared__ __attribute__((aligned(16))) C ring_lds[RING ? GPB * NH * (M / 2) : 1];      // per lane group: NH half-frames of M/2 complex
#ifndef FRT_WAVE_TABLES_IN_LDS        // experiment: one-wavefront float32 frames read their per-bin weights AND unpack twiddles from LDS
#define FRT_WAVE_TABLES_IN_LDS 0      // (17 registers fewer: four waves per SIMD without spills?)
#endif
    constexpr bool TLDS = FRT_WAVE_TABLES_IN_LDS && TPF == 64 && sizeof(T) == 4;
    constexpr bool WLDS = (RING && FRT_RING_WEIGHTS_IN_LDS) || TLDS;             // dB / colour-index offsets read from LDS per frame
    __shared__ T wgt_lds[WLDS ? M + 1 : 1];                                      // instead of nine registers held for the run
    __shared__ C twn_lds[TLDS ? M / 2 : 1];                                      // unpack twiddles of bins 0..M/2-1

    const int tid = threadIdx.x;
    // a lane group that is a whole wavefront: its index — and with it channel, run, frame range, row and sample bases — is
    // wave-uniform; said out loud, the compiler keeps that arithmetic (64-bit, ~22 instructions per frame) on the scalar unit
    // instead of the vector ALUs this kernel is bound by
    const int grp = TPF == 64 ? __builtin_amdgcn_readfirstlane(tid / TPF) : tid / TPF;
    const int i = tid - grp * TPF;
    C* buf = lds + grp * lds_padded_size(M);

    const int gg = blockIdx.x * GPB + grp;
    const bool group_ok = gg < a.n_groups;
    const int ggc = group_ok ? gg : 0;
    const int chan = ggc / a.runs_per_channel;
    const int run = ggc - chan * a.runs_per_channel;
    const long long f0 = a.frame_base + (long long)run * a.run;
    const long long left = a.n_frames - f0;
    int nfr = left > a.run ? a.run : (int)left;            // frames of this run (32-bit: the loop's compares stay scalar)
    if (!group_ok) nfr = 0;

    if (a.kind == FRT_STFT_IMAGE) {
        for (int t = threadIdx.x; t < 256; t += BLOCK) lut_lds[t] = a.lut[t];
        __syncthreads();
    }
    if constexpr (WLDS) {
        const T* wsrc = (const T*)(a.kind == FRT_STFT_IMAGE ? a.wimage : a.weight);
        for (int t = threadIdx.x; t <= M; t += BLOCK) wgt_lds[t] = wsrc ? wsrc[t] : (T)0;
        __syncthreads();
    }

    if constexpr (TLDS) {
        for (int t = threadIdx.x; t < M / 2; t += BLOCK) twn_lds[t] = ((const C*)a.twn)[t];
        __syncthreads();
    }

    const TIN* xc = (const TIN*)a.x + chan * a.x_stride;
    T* outc = (T*)a.out + chan * a.out_cstride;
    // complex point of the frame that register slot j of this thread holds before the transform
    auto point = [&](int j) -> int { return i + j * TPF; };

    // ---- per-thread constants ----------------------------------------------------------------
    // One-wave frames (N <= 1024) keep window, twiddles and weights in registers for the whole
    // run; many-wave frames re-read them from the (L2 resident) tables at every use because the
    // workgroup size caps their register budget (1024 threads -> 128 VGPRs).
    constexpr bool HOIST = WAVE;
    constexpr int NC = HOIST ? 8 : 1;
    C win[NC], twu[HOIST ? 4 : 1];
    T wdb[NC], wdb_mid = 0;
    const T* wtab = (const T*)a.window;
    const C* twn = (const C*)a.twn;
    // IMAGE folds dB, weighting, normalisation and the 255 of the LUT index into one multiply-add per
    // bin: index = clamp(gain * log2(P + 1e-30) + wimage[k], 0, 255) — same value as the reference's
    // clip((10 log10(P + eps) + w - min)/(max - min), 0, 1) * 255 up to float rounding
    const T* wgt = (const T*)(a.kind == FRT_STFT_IMAGE ? a.wimage : a.weight);
    const T image_gain = (T)a.image_gain;
    TwRegs<T, LOG2M> twr;
    if constexpr (HOIST) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = point(j);
            win[j] = {wtab[2 * n], wtab[2 * n + 1]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // bins i + j TPF and M - i - j TPF (see the unpack)
            if constexpr (!TLDS) twu[j] = twn[i + j * TPF];
            if constexpr (!WLDS) {
                wdb[j] = wgt ? wgt[i + j * TPF] : (T)0;
                wdb[4 + j] = wgt ? wgt[M - i - j * TPF] : (T)0;
            }
        }
        twr.load((const C*)a.tw, i);
    }
    if constexpr (!WLDS) {
        if (wgt) wdb_mid = wgt[M / 2];
    }

    const T norm_off = (T)a.norm_off, norm_scale = (T)a.norm_scale;
    static_assert(!SPLIT || WAVE, "split rows: N <= 1024");
    constexpr int PITCH = SPLIT ? M : M + 1;            // values per output row
    constexpr bool NYQ_REG = SPLIT && TPF == 64;        // lane g of these registers holds frame g's Nyquist value until the run ends
    uint32_t nyq_lo = 0, nyq_hi = 0;

    typedef T tv2 __attribute__((ext_vector_type(2)));     // a slot stays ONE 64-bit register pair from the load to the window multiply
    auto load_slot = [&](long long f, int j) -> tv2 {
        // frame base (wave-uniform for one-wavefront frames: a scalar pointer) + one 32-bit lane offset + the slot as an
        // immediate: the load's address costs no vector instruction
        const TIN* fb = xc + f * a.hop;
        const unsigned lane_off = 2u * (unsigned)point(0);
        const int slot_off = 2 * (point(j) - point(0));
        if (SHIFT != 0 || a.vec2) {         // the shifting instances are only launched on aligned 2-sample loads
#ifdef FRT_NT_LOADS
            typedef TIN vin2 __attribute__((ext_vector_type(2)));
            vin2 v = __builtin_nontemporal_load((const vin2*)(fb + lane_off + slot_off));
#else
            CIN v = *(const CIN*)(fb + lane_off + slot_off);
#endif
            return tv2{(T)v.x, (T)v.y};
        }
        return tv2{(T)fb[lane_off + slot_off], (T)fb[lane_off + slot_off + 1]};
    };

    // The register window.  A hop advances the frame by NEW of its 8 slots; the other 8 - NEW were loaded for earlier
    // frames.  The window is kept as NSETS = 8 / NEW physical sets of NEW slots whose ROLES rotate from frame to frame
    // (frame g finds its logical block b in physical set (b + g) mod NSETS) instead of their contents being moved: the
    // frame loop is unrolled NSETS times with the roles as compile-time constants, the set holding the oldest block is
    // refilled as soon as the window multiply has read it, and no register is ever copied (the copies were 16 of the
    // ~600 instructions a frame issues).
    constexpr int NEW = RING ? 4 : SHIFT == 0 ? 8 : SHIFT;       // slots fetched per frame
    constexpr int NSETS = RING ? NH : 8 / NEW;
    tv2 raw[RING ? 1 : 8];
    C* ring = ring_lds + (RING ? grp * NH * (M / 2) : 0);         // half-frame h of the run in slot h mod NH
    // half-frame h of the run (samples [h hop, (h + 1) hop) from the run's first frame) -> ring slot h mod NH: two wave-wide
    // 16-byte copies of 1 KB each
    auto ring_fetch = [&](long long h) {
        if constexpr (RING) {
            const char* src = (const char*)(xc + (f0 + h) * a.hop);
            const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(ring + (int)(h % NH) * (M / 2));
            const uint32_t lane16 = (uint32_t)i * 16;
#pragma unroll
            for (int part = 0; part < (int)(M * sizeof(T)) / 1024; ++part) {
                const unsigned long long ub = (unsigned long long)(src + part * 1024);
                const unsigned long long sb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :
                             : "v"(lane16), "s"(sb), "s"((uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + part * 1024)))
                             : "memory", "m0");
            }
        }
    };
    if constexpr (RING) {
        if (nfr > 0) {
            ring_fetch(0);
            ring_fetch(1);
            if (NH > 2 && nfr > 1) ring_fetch(2);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = tv2{(T)0, (T)0};
        if (nfr > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = load_slot(f0, j);
        }
    }

    // Drain the one-off loads (tables, first frame) here.  Without this the compiler's s_waitcnt
    // placement inside the loop has to assume they may still be in flight on the first trip and
    // emits vmcnt(0) at the top of every iteration, which would serialise the prefetch below.
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), other counters untouched

    const int nloop = a.run;                          // uniform trip count keeps barriers aligned
    // one frame; `phase` = g mod NSETS as a compile-time constant; false = the run is finished
    auto frame = [&](auto phase, const int g) -> bool {
        constexpr int PH = decltype(phase)::value;
        if (g >= nloop) return false;
        const bool valid = g < nfr;
        if (!WAVE) {
            if (!__syncthreads_or(valid)) return false;      // whole block finished
        } else if (!__any(valid)) {
            return false;
        }

        int zero = 0;
        if constexpr (!HOIST) asm volatile("s_mov_b32 %0, 0" : "=s"(zero));   // opaque per iteration
        // RING: the copy this frame's second half arrives by was issued a frame ago, in front of that frame's nine row
        // stores — the only younger vector-memory operations (the counter retires in order): at most nine outstanding
        // means the copy has landed.  (The first frame's two copies are drained before the loop.)
        // (split rows: EIGHT stores per frame — bin M/2 rides in lane 0's last descending store and the Nyquist bin waits in a
        // register for the end of the run; tests/test_ring_waitcnt.py counts both instances' stores in the generated code)
        if constexpr (RING) {
            static_assert(!SPLIT || NYQ_REG, "the ring instance counts its row stores");
            if constexpr (NH == 2) {
                if constexpr (SPLIT) __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8), other counters untouched
                else __builtin_amdgcn_s_waitcnt(0x0F79);                     // vmcnt(9)
            } else {
                // three halves: the copy was issued two frames ago; younger are two frames' stores and, if the previous
                // frame issued one, its copy (2 instructions)
                if constexpr (SPLIT) {
                    if (g + 1 < nfr) __builtin_amdgcn_s_waitcnt(0x4F72);      // vmcnt(18)
                    else __builtin_amdgcn_s_waitcnt(0x4F70);                  // vmcnt(16)
                } else {
                    if (g + 1 < nfr) __builtin_amdgcn_s_waitcnt(0x4F74);      // vmcnt(20)
                    else __builtin_amdgcn_s_waitcnt(0x4F72);                  // vmcnt(18)
                }
            }
        }

        C v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            C wj;
            if constexpr (HOIST) {
                wj = win[j];
            } else {
                wj = ((const C*)wtab)[i + j * TPF + zero];
            }
            C r;
            if constexpr (RING) {
                r = ring[((j / 4 + PH) % NH) * (M / 2) + point(j & 3)];          // first half: slot PH, second: the next
                // the sample itself is pinned, not the product: the multiply stays free to fuse into the first butterfly
                // exactly as in the register-window instance (the two instances give bit-identical spectra)
                asm volatile("" : "+v"(r.x), "+v"(r.y));
            } else {
                const tv2 rr = raw[((j / NEW + PH) % NSETS) * NEW + j % NEW];      // logical slot j of this frame
                r = {rr[0], rr[1]};
            }
            v[j] = {r.x * wj.x, r.y * wj.y};
        }
        if constexpr (RING) {
            // the eight LDS reads have returned (their values are pinned above): the first half's slot is free for
            // half-frame g + NH, which a later frame reads at least a whole transform from now
            asm volatile("" ::: "memory");
            if (g + NH - 1 < nfr) ring_fetch((long long)g + NH);
        } else {
        // The set that held the oldest block has been read: request the next frame's new slots into it now, a whole
        // transform ahead of their use.
        // (Unconditional, with the frame index clamped to the run's last frame: a conditional refill makes every slot a
        // merge of "old value" and "loaded value", which the register allocator resolves with copies at the loop's
        // back-edge — the very moves this scheme removes.  The one redundant request per run re-reads slots this wave
        // fetched a frame ago.)
        if constexpr (NSETS > 1) {
            int gn = g + 1 < nfr ? g + 1 : nfr - 1;
            if (gn < 0) gn = 0;
#ifdef FRT_ABLATE
            if (a.ablate & 2) gn = 0;
#endif
#pragma unroll
            for (int t = 0; t < NEW; ++t) raw[PH * NEW + t] = load_slot(f0 + gn, 8 - NEW + t);
        } else if (g + 1 < nfr) {       // a hop that reloads the whole frame: the redundant request would be a whole frame
#pragma unroll
            for (int t = 0; t < 8; ++t) raw[t] = load_slot(f0 + g + 1, t);
        }
        }

#ifdef FRT_ABLATE
        if (a.ablate & 4) {
        } else
#endif
        if constexpr (HOIST) {
            fft_pow2_forward<T, LOG2M, WAVE>(v, buf, i, twr);
        } else {
            TwTable<T, LOG2M> twt{(const C*)a.tw, zero};
            fft_pow2_forward<T, LOG2M, WAVE>(v, buf, i, twt);
        }

        // ---- conjugate-symmetric unpack, two bins at a time ------------------------------------------
        // With A = Z[k], B = conj Z[M-k], t = w^k (A - B):  X[k] = ((A+B) - i t)/2 and
        // X[M-k] = conj((A+B) + i t)/2, so one (A+B, t) serves both |X[k]|^2 and |X[M-k]|^2.  Thread i
        // finishes k = i + j TPF for j = 0..3 (k < M/2) together with M - k; k = 0 yields bins 0 and M,
        // and the self-paired bin M/2 is |Z[M/2]|^2 / N^2 (thread 0, slot 4).
        C part[4];
#ifdef FRT_ABLATE
        if (a.ablate & 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] = v[7 - j];
        } else
#endif
        if constexpr (WAVE) {
            const int lane = tid & 63;
            const int src = lane - i + ((TPF - i) & (TPF - 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // Z[M-k] sits in slot 7-j of lane TPF-i; lane 0 pairs inside itself: slot (8-j) mod 8
                const C mine = (i == 0) ? v[(8 - j) & 7] : v[7 - j];
                part[j] = {shfl_t(mine.x, src), shfl_t(mine.y, src)};
            }
        } else {
            __syncthreads();
#pragma unroll
            for (int j = 4; j < 8; ++j) buf[lds_pad(i + j * TPF)] = v[j];
            if (i == 0) buf[lds_pad(0)] = v[0];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] = buf[lds_pad((M - (i + j * TPF)) & (M - 1))];
        }

        T res[8];          // res[j] = P[i + j TPF], res[4 + j] = P[M - i - j TPF]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            C A = v[j], B = cconj(part[j]);
            C S = A + B, D = A - B;
            C tu;
            if constexpr (HOIST) {
                if constexpr (TLDS) tu = twn_lds[i + j * TPF];
                else tu = twu[j];
            } else {
                tu = twn[i + j * TPF + zero];
            }
            C t = cmul(tu, D);
            T ar = S.x + t.y, ai = S.y - t.x;      // 2 X[k]
            T br = S.x - t.y, bi = S.y + t.x;      // 2 conj X[M-k]
            res[j] = ar * ar + ai * ai;           // the window table carries the 1/(2N) scale (exact: a power of two)
            res[4 + j] = br * br + bi * bi;
        }
        T res_mid = (v[4].x * v[4].x + v[4].y * v[4].y) * (T)4;               // bin M/2, meaningful for i == 0

        // The prefetched slots are waited for HERE, in front of this frame's stores: the vector-memory counter retires in
        // order, so a first use behind the stores (the next frame's window multiply) could only be guarded by vmcnt(0) —
        // the acknowledgement of every row store.  The empty asm makes the loaded values a use at this point.
        if constexpr (!RING) {
#pragma unroll
            for (int t = 0; t < NEW; ++t) asm volatile("" : "+v"(raw[PH * NEW + t]));
        }

#ifdef FRT_ABLATE
        if (a.ablate & 1) {
            T acc = res_mid;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += res[j];
            if (acc == (T)-12345.678) outc[0] = acc;     // keeps the results live, never true
        } else
#endif
        if (valid) {
            // element offsets of the 8 (+1) bins inside the row
            const int klo = i, khi = M - i;
            T* row = outc + (f0 + g) * PITCH;
            // the frame's nine values: vals[j] = bin klo + j TPF, vals[4 + j] = bin khi - j TPF, mid = bin M/2 (lane 0 of the group)
            auto store_row = [&](auto* r, const auto* vals, auto mid) {
                if constexpr (!SPLIT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        stream_store(r + klo + j * TPF, vals[j]);
                        stream_store(r + khi - j * TPF, vals[4 + j]);
                    }
                    if (i == 0) stream_store(r + M / 2, mid);
                } else {
                    // descending side: lanes 1..TPF-1 hold bins M - j TPF - i; lane 0 holds M - j TPF, which belongs to the
                    // 64-value block ABOVE — it stores the value it holds for that block's lowest bin instead (slot j + 1, or
                    // the self-paired bin M/2 for j = 3), so that every store covers [M - (j + 1) TPF, M - j TPF) exactly
                    const int ihi = i == 0 ? TPF : i;
                    typedef std::remove_cv_t<std::remove_pointer_t<decltype(vals)>> V;
                    // non-temporal: every store of a one-wavefront frame writes whole 64-byte lines here, which the L2 can pass on
                    // without keeping them (measured on aligned buffer sets, profiles/r05_headline_variants.txt: colour kind 0.117 ->
                    // 0.113 ms, PSD kind 0.109 -> 0.105; on the packed rows' partial lines the same hint costs 3 %)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        auto hv = vals[4 + j];
                        if (i == 0) hv = j < 3 ? vals[5 + (j < 3 ? j : 0)] : mid;
#ifdef FRT_SPLIT_PLAIN_STORES
                        stream_store(r + klo + j * TPF, vals[j]);
                        stream_store(r + M - ihi - j * TPF, hv);
#else
                        __builtin_nontemporal_store(vals[j], r + klo + j * TPF);
                        __builtin_nontemporal_store(hv, r + M - ihi - j * TPF);
#endif
                    }
                    // bin M (lane 0, slot 4)
                    if constexpr (NYQ_REG) {
                        nyq_lo = lane_insert(nyq_lo, value_bits_lo(vals[4]), g & 63, i);
                        if constexpr (sizeof(*vals) == 8) nyq_hi = lane_insert(nyq_hi, value_bits_hi(vals[4]), g & 63, i);
                    } else if (i == 0) {
                        stream_store((V*)a.out_nyq + chan * a.n_frames + (f0 + g), vals[4]);
                    }
                }
            };
            if (a.kind == FRT_STFT_PSD) {
                store_row(row, res, res_mid);
            } else {
                T wl[4], wh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (WLDS) {
                        wl[j] = wgt_lds[klo + j * TPF];
                        wh[j] = wgt_lds[khi - j * TPF];
                    } else if constexpr (HOIST) {
                        wl[j] = wdb[j];
                        wh[j] = wdb[4 + j];
                    } else {
                        wl[j] = wgt ? wgt[klo + j * TPF + zero] : (T)0;
                        wh[j] = wgt ? wgt[khi - j * TPF + zero] : (T)0;
                    }
                }
                if constexpr (WLDS) wdb_mid = wgt_lds[M / 2];
                if (a.kind == FRT_STFT_IMAGE) {
                    // colour words are 4 bytes whatever the arithmetic type
                    uint32_t* prow = (uint32_t*)a.out + chan * a.out_cstride + (f0 + g) * PITCH;
                    // float32: with the dB floor below the LUT's range everywhere, P + 1e-30 rounds to P for every P that
                    // is not clamped to index 0 anyway, and the add is left out (eps_free, frt_stft_set_epilogue)
                    auto colour_row = [&](auto eps_free) {
                        constexpr bool EPS_FREE = decltype(eps_free)::value;
                        // float64 instance: the same float32 evaluation of the index from the power rounded to float32 (its own
                        // 2^-24 is inside `thr`), the float64 comparison for the few bins next to an edge — no float64 logarithm
                        // (software, ~40 instructions) per bin
                        const float gain32 = (float)image_gain;
                        auto index_value = [&](T pw, T w) -> float {
                            float wf;
                            if constexpr (sizeof(T) == 8) wf = (float)(w + (T)a.image_thr);
                            else wf = (float)w;
                            if constexpr (EPS_FREE) return clamp_index(gain32 * __log2f((float)pw) + wf);
                            else return clamp_index(gain32 * __log2f((float)pw + 1e-30f) + wf);
                        };
                        float q[9];
                        uint32_t colour[9];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            q[j] = index_value(res[j], wl[j]);
                            q[4 + j] = index_value(res[4 + j], wh[j]);
                        }
                        q[8] = index_value(res_mid, wdb_mid);               // stored by thread 0 only
#pragma unroll
                        for (int j = 0; j < 9; ++j) colour[j] = lut_lds[(int)q[j]];
                        {
                            // Bins within 2 thr above an index edge (3e-4 of them) are decided in float64 (exact_colour_index).
                            // The LUT reads above are issued first, with the float32 index, so that the frame's only branch
                            // sits behind them and in front of nothing but the stores; the rare path recomputes what it
                            // needs from the powers (nothing but those nine values and the colours stays live across it).
                            float fmin9 = i == 0 ? __builtin_amdgcn_fractf(q[8]) : 1.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) fmin9 = fminf(fmin9, __builtin_amdgcn_fractf(q[j]));
                            if (__any(fmin9 < a.edge2)) {
                                // ONE instance of the float64 decision: every lane with such a bin picks its first one
                                // (select chains over the nine register slots), the wave serves those lanes one after the
                                // other (exact_colour_index), and the colour goes back through a select chain; a lane with
                                // two such bins in one frame (1e-5 of the frames) goes round again.
                                // (the index values are recomputed from the powers here — bit-identical, same operations —
                                // so that only the nine powers, not the nine index values as well, stay live across the branch)
                                uint32_t pend = 0;
                                auto power = [&](int j) -> T { return j < 8 ? res[j] : res_mid; };
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    pend |= (__builtin_amdgcn_fractf(index_value(power(j), wl[j])) < a.edge2 ? 1u : 0u) << j;
                                    pend |= (__builtin_amdgcn_fractf(index_value(power(4 + j), wh[j])) < a.edge2 ? 1u : 0u) << (4 + j);
                                }
                                if (i == 0) pend |= (__builtin_amdgcn_fractf(index_value(power(8), wdb_mid)) < a.edge2 ? 1u : 0u) << 8;
#pragma unroll 1
                                while (__any(pend != 0)) {
                                    const int jsel = pend ? __ffs(pend) - 1 : 0;
                                    T psel = power(8), wsel = wdb_mid;
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        psel = jsel == j ? res[j] : jsel == 4 + j ? res[4 + j] : psel;
                                        wsel = jsel == j ? wl[j] : jsel == 4 + j ? wh[j] : wsel;
                                    }
                                    const float qsel = index_value(psel, wsel);
                                    const int ksel = jsel < 4 ? klo + jsel * TPF : jsel < 8 ? khi - (jsel - 4) * TPF : M / 2;
                                    const uint32_t c = lut_lds[exact_colour_index(pend != 0, psel, ksel, (int)qsel, a)];
#pragma unroll
                                    for (int j = 0; j < 9; ++j) colour[j] = (pend != 0 && jsel == j) ? c : colour[j];
                                    pend &= pend - 1;
                                }
                            }
                        }
                        store_row(prow, colour, colour[8]);
                    };
                    if (a.eps_free) colour_row(std::true_type{});
                    else colour_row(std::false_type{});
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        res[j] = db10<T>(res[j]) + wl[j];
                        res[4 + j] = db10<T>(res[4 + j]) + wh[j];
                    }
                    res_mid = db10<T>(res_mid) + wdb_mid;
                    if (a.kind == FRT_STFT_NORM) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) res[j] = (res[j] + norm_off) * norm_scale;
                        res_mid = (res_mid + norm_off) * norm_scale;
                    }
                    store_row(row, res, res_mid);
                }
            }
        }
        return true;
    };
    for (int g = 0; g < nloop; g += NSETS) {
        if (!frame(std::integral_constant<int, 0>{}, g)) break;
        if constexpr (NSETS > 1) {
            if (!frame(std::integral_constant<int, 1>{}, g + 1)) break;
        }
        if constexpr (NSETS > 2) {
            if (!frame(std::integral_constant<int, 2>{}, g + 2)) break;
            if constexpr (NSETS > 3) {
                if (!frame(std::integral_constant<int, 3>{}, g + 3)) break;
            }
        }
    }
    if constexpr (NYQ_REG) {
        // the run's Nyquist values: lane g holds frame g's (runs are at most 64 frames, stft_launch)
        if (i < nfr) {
            const long long at = chan * a.n_frames + f0 + i;
            if (a.kind == FRT_STFT_IMAGE || sizeof(T) == 4) ((uint32_t*)a.out_nyq)[at] = nyq_lo;
            else ((double*)a.out_nyq)[at] = __hiloint2double((int)nyq_hi, (int)nyq_lo);
        }
    }
}

}  // namespace frt
