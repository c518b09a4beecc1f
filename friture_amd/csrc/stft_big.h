// stft_big.h — the instances of K1 for N >= 2048 (a frame spans several wavefronts), float32.
// Included by stft.hip.
//
// The generic stft_kernel runs five Stockham passes over the whole workgroup for N = 16384: ten
// workgroup barriers per frame, every pass re-reading its twiddles from L2 between barriers, one
// 1024-thread workgroup per CU with nothing to overlap the stalls — 14 % of HBM peak.  This instance
// factors M = N/2 = 16 * Ms so that only ONE exchange crosses wavefronts:
//   1. thread t (Ms threads per frame) loads z[t + j Ms], j = 0..15 (coalesced), windows them and
//      takes the 16-point DFT over j in registers;
//   2. multiplies by exp(-2 pi i t k0 / M) and writes column k0 to LDS region k0 (the only
//      workgroup-wide transpose), barrier;
//   3. the 16 length-Ms transforms over t run as wave-local sub-transforms (fft_core.h, no barrier:
//      Ms <= 512 points = one lane group of Ms/8 threads inside a wavefront), two rounds of eight;
//      results return to their LDS region, barrier;
//   4. thread t finishes the bin pairs (k, M-k), k = t + q Ms (q < 8), reading Z[k] = region
//      (k mod 16), slot (k / 16): consecutive threads -> consecutive bins -> coalesced stores.
// Three barriers per frame, 74 KB of LDS for N = 16384 -> two 512-thread workgroups per CU.
#pragma once

#ifndef FRT_BIG_DMA_MIN_LOG2M    // smallest log2(N/2) whose aligned launches take the LDS-staged instance
#define FRT_BIG_DMA_MIN_LOG2M 11
#endif
#ifndef FRT_BIG_UNROLL_ROUNDS    // 1: both rounds of sub-transforms as straight-line code (LDS addresses of the second round as immediate offsets: +1-2 %)
#define FRT_BIG_UNROLL_ROUNDS 1
#endif
#ifndef FRT_BIG_ZB              // bin pairs whose Z values are requested together in the unpack (2, 4 or 8)
#define FRT_BIG_ZB 4
#endif
#ifndef FRT_BIG_ABLATE          // experiment builds only: 1 no row stores, 2 no sample loads after the first frame, 4 no sub-transforms
#define FRT_BIG_ABLATE 0
#endif

namespace frt {

// Second half of a 16-point DFT.  With n = m + 4p and k = q + 4r,
//   X[q + 4r] = sum_m W4^{mr} ( W16^{mq} sum_p a[m + 4p] W4^{pq} );
// on entry a[m + 4q] holds the inner sum (first-stage butterfly m, output q), on return a[k] = X[k].
template <typename T>
__device__ __forceinline__ void dft16_second_stage(cpx<T> (&a)[16]) {
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;   // cos, sin(pi/8)
    const T h = (T)0.70710678118654752440;
    auto mulc = [](cpx<T> v, T c, T s) -> cpx<T> { return {v.x * c + v.y * s, v.y * c - v.x * s}; };   // v (c - i s)
    // twiddles W16^{mq} on a[m + 4q]
    a[1 + 4] = mulc(a[1 + 4], c1, s1);        // m=1,q=1: W16^1
    a[1 + 8] = mulc(a[1 + 8], h, h);          // m=1,q=2: W16^2
    a[1 + 12] = mulc(a[1 + 12], s1, c1);      // m=1,q=3: W16^3
    a[2 + 4] = mulc(a[2 + 4], h, h);          // W16^2
    a[2 + 8] = mul_mi(a[2 + 8]);              // W16^4 = -i
    a[2 + 12] = mulc(a[2 + 12], -h, h);       // W16^6
    a[3 + 4] = mulc(a[3 + 4], s1, c1);        // W16^3
    a[3 + 8] = mulc(a[3 + 8], -h, h);         // W16^6
    a[3 + 12] = mulc(a[3 + 12], -c1, -s1);    // W16^9 = -W16^1
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // DFT4 over m of a[m + 4q] -> X[q + 4r] at r = 0..3; stored back into the four slots of column q
        dft4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
    }
    // a[4q + r] = X[q + 4r]: transpose the 4x4 index grid into natural order
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = q + 1; r < 4; ++r) {
            const cpx<T> tmp = a[4 * q + r];
            a[4 * q + r] = a[4 * r + q];
            a[4 * r + q] = tmp;
        }
}

template <int LOG2M>
struct BigPlan {
    static constexpr int M = 1 << LOG2M;
    static constexpr int LOG2MS = LOG2M - 4;
    static constexpr int MS = M / 16;                       // threads per frame = sub-transform length
    static constexpr int TPFS = MS / 8;                     // threads per sub-transform
    static constexpr int RS = lds_padded_size(MS) + 2;      // region stride (complex): conflict-free column reads
    // One frame per workgroup.  (Workgroups of 256 threads serving 4 / 2 frames at N = 2048 / 4096 tied unrelated
    // frames to the same three barriers per frame: a wave held up behind its row stores held up three others — 19 %
    // at N = 2048.  At N = 2048 the workgroup is one wavefront and the barriers compile to nothing.)
    static constexpr int BLOCK = MS;
    static constexpr int GPB = 1;                           // frames in flight per workgroup
};

// DMA (N >= 4096, 16-byte aligned rows): the samples of the NEXT frame are copied from HBM
// straight into LDS (global_load_lds_dwordx4, no registers involved) while the current frame's sub-transforms and
// unpack run; the first stage then reads its 16 samples from LDS.  These sizes cannot afford the 32 live registers a
// register prefetch costs (they hold ~90 hoisted constants), and have the LDS to spare.
// T: float (every optimisation below) or double (the transform structure only: no register-resident constants, no
// prefetch, no staging — the float64 instance serves the drop-in / pitch-tracker path, twice the registers and LDS).
template <typename T, int LOG2M, bool DMA>
__device__ __forceinline__ void stft_big_body(const StftArgs& a, cpx<T>* __restrict__ lds, uint32_t* __restrict__ lut_lds,
                                              cpx<T>* __restrict__ stage, cpx<T>* __restrict__ tws_lds) {
    // (The LDS arrays arrive as __restrict__ parameters of an inlined function so that every LDS access carries an alias
    // scope.  That alone did not keep the compiler's wait-count pass from guarding LDS accesses behind an LDS-DMA with
    // vmcnt(0) — paired ds_write2 lose their scopes — which is why the copy is issued from inline assembly, below.)
    using B = BigPlan<LOG2M>;
    using C = cpx<T>;
    constexpr int M = B::M, MS = B::MS, TPFS = B::TPFS, RS = B::RS, BLOCK = B::BLOCK;
    static_assert(B::GPB == 1 && BLOCK == MS, "one frame per workgroup");
    constexpr bool TWLDS = sizeof(T) == 8;

    const int tid = threadIdx.x;
    const int t = tid;
    C* reg = lds;

    if (a.kind == FRT_STFT_IMAGE) {
        for (int q = tid; q < 256; q += BLOCK) lut_lds[q] = a.lut[q];
    }
    if constexpr (TWLDS) {
        for (int q = tid; q < MS; q += BLOCK) tws_lds[q] = ((const C*)a.tws)[q];      // visible after the loop's first barrier
    }

    // everything that places the workgroup is uniform: scalar registers, scalar-base addressing for loads and stores
    const int gg = blockIdx.x;
    const int chan = gg / a.runs_per_channel;
    const int run = gg - chan * a.runs_per_channel;
    const long long f0 = a.frame_base + (long long)run * a.run;
    int nfr = (int)(a.n_frames - f0 < (long long)a.run ? a.n_frames - f0 : (long long)a.run);
    if (gg >= a.n_groups || nfr < 0) nfr = 0;

    const C* xs = (const C*)((const T*)a.x + chan * a.x_stride);
    const C* win = (const C*)a.window;
    const C* tw = (const C*)a.tw;          // exp(-2 pi i n / M)
    const C* twn = (const C*)a.twn;        // exp(-2 pi i k / N)
    const T* wgt = (const T*)(a.kind == FRT_STFT_IMAGE ? a.wimage : a.weight);
    const T image_gain = (T)a.image_gain, norm_off = (T)a.norm_off, norm_scale = (T)a.norm_scale;

    // twiddles of the wave-local sub-transforms depend on the thread's index in its sub-transform only
    const int si = t % TPFS;               // index inside the sub-transform
    const int sg = t / TPFS;               // sub-transform of a round (0..7)
    // float64: the sub-transform twiddles are re-read from the table (56 registers otherwise: one wave per SIMD)
    constexpr bool TWREG = sizeof(T) == 4;
    TwRegs<T, TWREG ? B::LOG2MS : 3> twr;
    if constexpr (TWREG) twr.load((const C*)a.tws, si);
    // so do the thread's other per-frame constants — the 15 factors exp(-2 pi i t k0 / M) between the radix-16
    // stage and the sub-transforms, its 16 window pairs, its 8 unpack factors and the 16 dB / colour-index offsets
    // of its bins: ~90 registers (239 in all at N = 16384) instead of ~55 L2 reads per frame.  Measured with runs of
    // 8-16 frames: +37 % at N = 16384 (one 512-thread workgroup per CU either way), +9...15 % at 4096 / 8192 (two
    // 256-thread workgroups per CU instead of three), +3 % at 2048.
    constexpr bool HOIST1 = LOG2M >= 10 && sizeof(T) == 4;
    C tw1[HOIST1 ? 15 : 1];
    C winr[HOIST1 ? 16 : 1];                    // the thread's 16 window pairs, same condition
    C twur[HOIST1 ? 8 : 1];                     // and the 8 unpack factors exp(-2 pi i k / N), k = t + q Ms
    // ... and the dB / colour-index offsets of the thread's 16 bins
    constexpr bool HOISTW = HOIST1;
    T wgr[HOISTW ? 16 : 1];
    const T wg_nyq = (wgt && a.kind != FRT_STFT_PSD) ? wgt[M / 2] : (T)0;      // weight of bin N/2 (thread 0 stores it)
    if constexpr (HOIST1) {
#pragma unroll
        for (int k0 = 1; k0 < 16; ++k0) tw1[k0 - 1] = tw[(t * k0) & (M - 1)];
#pragma unroll
        for (int j = 0; j < 16; ++j) winr[j] = win[t + j * MS];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            twur[q] = twn[t + q * MS];
            if constexpr (HOISTW) {
                wgr[q] = wgt ? wgt[t + q * MS] : (T)0;
                wgr[8 + q] = wgt ? wgt[M - t - q * MS] : (T)0;
            }
        }
    }

    // z[t + j Ms] of a frame: scalar base (the frame's first sample + j Ms, all uniform) + ONE lane offset, 8 t bytes —
    // spelled out, because the compiler's own split kept several 64-bit per-lane pointers alive across the frame loop
    // (`opaque0`: a zero the compiler cannot see through, defined inside the frame loop — it keeps "row start + 8 t" from
    // being hoisted out of the loop as a 64-bit per-lane pointer again)
    const uint32_t t_bytes = (uint32_t)t * (uint32_t)sizeof(C);
    auto sample = [&](long long frame, int j, int opaque0 = 0) -> C {
        const unsigned long long ub = (unsigned long long)(xs + (frame * a.hop >> 1) + j * MS);
        // (readfirstlane of a uniform value is free, and pins the base to scalar registers: scalar-base addressing)
        const unsigned long long sb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
        return *(const C*)((const char*)sb + (t_bytes + (uint32_t)opaque0));
    };
    // N <= 4096: the 16 samples of the NEXT frame are requested as soon as the current frame's first stage has left
    // its registers (32 more live registers: +15 % at 2048 / 4096; at 8192 the kernel would drop to one wave per SIMD
    // or lose the hoisted weights, at 16384 it spills — measured slower or equal there)
    constexpr bool PREFETCH = HOIST1 && !DMA && LOG2M <= 11;
    C nx[PREFETCH ? 16 : 1];                     // the samples of the frame about to be transformed
    if constexpr (PREFETCH) {
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = nfr > 0 ? sample(f0, j) : C{(T)0, (T)0};
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = C{nx[j].x * winr[j].x, nx[j].y * winr[j].y};      // nx holds WINDOWED samples
    }
    // one frame = M complex = 8 M bytes = M / 128 wave-instructions of 1 KB, dealt round-robin to the wavefronts.
    // Issued from inline assembly (scalar row base + lane offset, LDS address in M0): an LDS-DMA the compiler knows of
    // makes its wait-count pass guard LDS accesses behind it with vmcnt(0) wherever it cannot prove them disjoint from
    // the staging buffer — it did so at the first LDS access of the sub-transforms, i.e. the copy of the next frame was
    // awaited right where it had been issued and overlapped with nothing.  The waits are placed by hand instead: the
    // vmcnt(16) at the top of the frame loop (vector-memory operations retire in order, and the only ones younger than
    // the copy are the frame's row stores).
    const uint32_t stage_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)stage;
    auto stage_frame = [&](long long frame) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const char* src = (const char*)(xs + (frame * a.hop >> 1)) + wave * 1024;
        const uint32_t dst = stage_lds + wave * 1024;
        const uint32_t lane16 = (tid & 63) * 16;
#pragma unroll
        for (int j = 0; j < M / 128 / (BLOCK / 64); ++j) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :
                         : "v"(lane16), "s"(src + j * (BLOCK / 64) * 1024), "s"(dst + j * (BLOCK / 64) * 1024)
                         : "memory", "m0");
        }
    };
    if constexpr (DMA) {
        if (nfr > 0) stage_frame(f0);
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): nothing younger is in flight yet that the loop's vmcnt(16) could count on
    }
    for (int g = 0; g < nfr; ++g) {                           // nfr is uniform over the workgroup
        // the staged frame has landed once at most the 16 row stores of the previous frame are still outstanding
        if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x4F70);           // vmcnt(16), other counters untouched
        __syncthreads();                                      // fences the previous frame's LDS reads
        int zero = 0;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));        // keeps table loads inside the loop

        // ---- 1. load + window + 16-point DFT over j -------------------------------------------------------
        // n = t + (m + 4p) Ms: the four points of first-stage butterfly m arrive together; loads run one
        // butterfly ahead of the arithmetic so that at most two groups of samples + window are in flight
        // (all sixteen at once would need 64 more registers and halve the occupancy)
        C v[16];
        if constexpr (PREFETCH) {
            // the frame's 16 samples were requested during the previous frame's sub-transforms (or before the loop)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                C b0 = nx[m], b1 = nx[m + 4], b2 = nx[m + 8], b3 = nx[m + 12];       // already windowed
                dft4(b0, b1, b2, b3);
                v[m] = b0; v[m + 4] = b1; v[m + 8] = b2; v[m + 12] = b3;      // v[m + 4q] = first-stage output q of butterfly m
            }
        } else if constexpr (DMA) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                C d[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) d[p] = stage[t + (m + 4 * p) * MS];
                C b0 = {d[0].x * winr[m].x, d[0].y * winr[m].y};
                C b1 = {d[1].x * winr[m + 4].x, d[1].y * winr[m + 4].y};
                C b2 = {d[2].x * winr[m + 8].x, d[2].y * winr[m + 8].y};
                C b3 = {d[3].x * winr[m + 12].x, d[3].y * winr[m + 12].y};
                dft4(b0, b1, b2, b3);
                v[m] = b0; v[m + 4] = b1; v[m + 8] = b2; v[m + 12] = b3;
            }
        } else {
            const C* wf = win + t + zero;
            C d[4], w[4], dn[4], wn[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                d[p] = sample(f0 + g, 4 * p, zero);
                if constexpr (HOIST1) w[p] = winr[4 * p];
                else w[p] = wf[(4 * p) * MS];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < 3) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        dn[p] = sample(f0 + g, m + 1 + 4 * p, zero);
                        if constexpr (HOIST1) wn[p] = winr[m + 1 + 4 * p];
                        else wn[p] = wf[(m + 1 + 4 * p) * MS];
                    }
                }
                C b0 = {d[0].x * w[0].x, d[0].y * w[0].y}, b1 = {d[1].x * w[1].x, d[1].y * w[1].y};
                C b2 = {d[2].x * w[2].x, d[2].y * w[2].y}, b3 = {d[3].x * w[3].x, d[3].y * w[3].y};
                dft4(b0, b1, b2, b3);
                v[m] = b0; v[m + 4] = b1; v[m + 8] = b2; v[m + 12] = b3;      // v[m + 4q] = first-stage output q of butterfly m
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 4; ++p) { d[p] = dn[p]; w[p] = wn[p]; }
            }
        }
        dft16_second_stage(v);
        // ---- 2. twiddle exp(-2 pi i t k0 / M), transpose through LDS ----------------------------------------
#pragma unroll
        for (int k0 = 1; k0 < 16; ++k0) {
            if constexpr (HOIST1) v[k0] = cmul(v[k0], tw1[k0 - 1]);
            else v[k0] = cmul(v[k0], tw[((t * k0) & (M - 1)) + zero]);
        }
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) reg[k0 * RS + lds_pad(t)] = v[k0];
        if constexpr (PREFETCH) {
            // v is dead from here on: request the next frame's samples now, a whole round of sub-transforms and the
            // unpack ahead of their use (the barriers below only wait for LDS traffic)
            if ((FRT_BIG_ABLATE & 2) && a.n_frames > 0) {
                // keep the first frame's samples
            } else if (g + 1 < nfr) {
#pragma unroll
                for (int j = 0; j < 16; ++j) nx[j] = sample(f0 + g + 1, j, zero);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) nx[j] = C{(T)0, (T)0};
            }
        }
        __syncthreads();
        if constexpr (DMA) {
            // every thread has read its samples of this frame: the staging buffer is free for the next one
            if (g + 1 < nfr && !((FRT_BIG_ABLATE & 2) && a.n_frames > 0)) stage_frame(f0 + g + 1);
        }
        // ---- 3. sixteen wave-local transforms of length Ms over t, two rounds of eight ----------------------
#if FRT_BIG_UNROLL_ROUNDS
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int r = 0; r < ((FRT_BIG_ABLATE & 4) ? (a.n_frames < 0 ? 2 : 0) : 2); ++r) {
            C* buf = reg + (8 * r + sg) * RS;
            C u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = buf[lds_pad(si + j * TPFS)];
            if constexpr (TWREG) {
                fft_pow2_forward<T, B::LOG2MS, true>(u, buf, si, twr);
            } else {
                TwTable<T, B::LOG2MS> twt{tws_lds, zero};
                fft_pow2_forward<T, B::LOG2MS, true>(u, buf, si, twt);
            }
            pass_sync<true>();
#pragma unroll
            for (int j = 0; j < 8; ++j) buf[lds_pad(si + j * TPFS)] = u[j];
        }
        __syncthreads();
        // ---- 4. conjugate-symmetric unpack of the pairs (k, M - k), k = t + q Ms ------------------------------
        auto zat = [&](int k) -> C {                         // Z[k], k in [0, M]
            k &= M - 1;
            return reg[(k & 15) * RS + lds_pad(k >> 4)];
        };
        // The Z values of ZB of the thread's eight pairs are requested before the first is used (the registers of the
        // first stage are free here): 8 / ZB exposed LDS round trips per frame instead of eight — with two waves per
        // SIMD there is little else to cover them.  (All eight at once spill where the prefetched samples are live too.)
        constexpr int ZB = (PREFETCH && LOG2M == 11) ? 2 : FRT_BIG_ZB;      // (the unaligned-row fallback at N = 4096 has no registers left)
        C za[8], zb[8];
        auto request_z = [&](int q0) {
#pragma unroll
            for (int q = q0; q < q0 + ZB; ++q) {
                za[q] = zat(t + q * MS);
                zb[q] = zat(M - t - q * MS);
            }
        };
        request_z(0);
        // (without the scheduling barrier the batches are merged again, and the kernel spills)
        auto next_z = [&](int q0) {
            __builtin_amdgcn_sched_barrier(0);
            request_z(q0);
        };
        if constexpr (PREFETCH) {
            // The next frame's samples are windowed HERE, before this frame's row stores are issued.  The vector-memory
            // counter retires in order: a first use after the stores (the top of the next iteration) can only be
            // guarded by vmcnt(0) = wait for the acknowledgement of every store of every frame (measured: the stores
            // then cost 44 % of the kernel's time at N = 2048).  Here vmcnt(0) covers loads that have had both
            // sub-transform rounds to arrive, plus stores that are a whole frame old.
#pragma unroll
            for (int j2 = 0; j2 < 16; ++j2) {
                const C w = winr[j2];
                nx[j2] = C{nx[j2].x * w.x, nx[j2].y * w.y};
                asm volatile("" : "+v"(nx[j2].x), "+v"(nx[j2].y));      // the products exist here: not sunk to their use
            }
        }
        {
            T* row = (T*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);
            // colour words are 4 bytes whatever the arithmetic type
            uint32_t* prow = (uint32_t*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);
            constexpr bool NOSTORE = (FRT_BIG_ABLATE & 1) != 0;
            // Row stores: scalar base (row + the pair's offset) + one of two lane offsets, 4 tz for the bins t + q Ms and
            // 4 (Ms - tz) for the bins M - t - q Ms.  tz = t + an opaque 0: otherwise the sixteen lane offsets are
            // hoisted out of the frame loop into sixteen registers the kernel does not have.
            const uint32_t tz = (uint32_t)(t + zero), tzh = (uint32_t)(MS - t - zero);
            auto lo_at = [&](auto* base, int q) { return base + q * MS + tz; };                    // bin t + q Ms
            auto hi_at = [&](auto* base, int q) { return base + (M - MS - q * MS) + tzh; };        // bin M - t - q Ms
            auto finish_store = [&](T* dst, uint32_t* pdst, int k, T p, T w) {      // the dB kinds (and the float64 colour index)
                if (NOSTORE && p != (T)-1) return;                // never true for a power: the arithmetic stays
                if (a.kind == FRT_STFT_IMAGE) {
                    const T vv = clamp_index(image_gain * log2_t(p + (T)1e-30) + w);
                    int idx = (int)vv;
                    if constexpr (sizeof(T) == 4) {
                        // within 2 thr above an index edge: one float64 comparison decides (stft.hip, exact_colour_index)
                        const bool near_edge = __builtin_amdgcn_fractf(vv) < a.edge2;
                        if (__any(near_edge)) idx = exact_colour_index(near_edge, p, k, idx, a);
                    }
                    *pdst = lut_lds[idx];                            // colour words are 4 bytes whatever T is
                } else {
                    T vv = db10<T>(p) + w;
                    if (a.kind == FRT_STFT_NORM) vv = (vv + norm_off) * norm_scale;
                    *dst = vv;
                }
            };
            auto weight_at = [&](int k) -> T { return wgt ? wgt[k + zero] : (T)0; };
            auto pair_powers = [&](int q, C wk, T& plo, T& phi) {
                const C A = za[q], Bc = cconj(zb[q]);
                const C S = A + Bc, D = A - Bc;
                const C tt = cmul(wk, D);
                const T ar = S.x + tt.y, ai = S.y - tt.x, br = S.x - tt.y, bi = S.y + tt.x;
                plo = ar * ar + ai * ai;
                phi = br * br + bi * bi;
            };
            // IMAGE kind, float32: four bins at a time — index values, LUT reads (issued with the float32 index), ONE
            // near-edge test for the four, stores.  A test per bin put a branch, and behind it an exposed LUT read,
            // between every two of the thread's 17 stores (+8 % on the kernel).
            auto image4 = [&](int q, const int (&k)[4], const T (&pw)[4], const T (&w)[4]) {
                T vv[4];
                uint32_t c[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vv[e] = clamp_index(image_gain * log2_t(pw[e] + (T)1e-30) + w[e]);
                    c[e] = lut_lds[(int)vv[e]];
                }
                if constexpr (sizeof(T) == 4) {
                    const float m = fminf(fminf(__builtin_amdgcn_fractf(vv[0]), __builtin_amdgcn_fractf(vv[1])),
                                          fminf(__builtin_amdgcn_fractf(vv[2]), __builtin_amdgcn_fractf(vv[3])));
                    if (__any(m < a.edge2)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool near_edge = __builtin_amdgcn_fractf(vv[e]) < a.edge2;
                            const int n = exact_colour_index(near_edge, pw[e], k[e], (int)vv[e], a);
                            if (near_edge) c[e] = lut_lds[n];
                        }
                    }
                }
                *lo_at(prow, q) = c[0];
                *hi_at(prow, q) = c[1];
                *lo_at(prow, q + 1) = c[2];
                *hi_at(prow, q + 1) = c[3];
            };
            // unpack factors and weights: registers (float32), or requested here — ALL of them before the first row
            // store: a table load issued between the stores of two bin pairs is only complete, for the in-order
            // vector-memory counter, once those stores are acknowledged (float64)
            C twl[8];
            T wl[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if constexpr (HOIST1) twl[q] = twur[q];
                else twl[q] = twn[t + q * MS + zero];
            }
            if (a.kind != FRT_STFT_PSD) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (HOISTW) {
                        wl[q] = wgr[q];
                        wl[8 + q] = wgr[8 + q];
                    } else {
                        wl[q] = weight_at(t + q * MS);
                        wl[8 + q] = weight_at(M - t - q * MS);
                    }
                }
            }
            // the output kind is uniform: one branch per frame, not one per bin
            if (a.kind == FRT_STFT_PSD) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    T plo, phi;
                    if (q > 0 && q % ZB == 0) next_z(q);
                    pair_powers(q, twl[q], plo, phi);
                    if (!NOSTORE || plo == (T)-1) {
                        *lo_at(row, q) = plo;
                        *hi_at(row, q) = phi;
                    }
                }
            } else if (sizeof(T) == 4 && a.kind == FRT_STFT_IMAGE && !NOSTORE) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    T pw[4];
                    if (q > 0 && q % ZB == 0) next_z(q);
                    pair_powers(q, twl[q], pw[0], pw[1]);
                    pair_powers(q + 1, twl[q + 1], pw[2], pw[3]);
                    const int k[4] = {t + q * MS, M - t - q * MS, t + (q + 1) * MS, M - t - (q + 1) * MS};
                    const T w[4] = {wl[q], wl[8 + q], wl[q + 1], wl[8 + q + 1]};
                    image4(q, k, pw, w);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    T plo, phi;
                    if (q > 0 && q % ZB == 0) next_z(q);
                    pair_powers(q, twl[q], plo, phi);
                    finish_store(lo_at(row, q), lo_at(prow, q), t + q * MS, plo, wl[q]);
                    finish_store(hi_at(row, q), hi_at(prow, q), M - t - q * MS, phi, wl[8 + q]);
                }
            }
            if (t == 0) {
                const C zm = zat(M / 2);
                const T pm = (zm.x * zm.x + zm.y * zm.y) * (T)4;
                // (a weight load here, after the row stores, would put vmcnt(0) at the loop top: wg_nyq is a register)
                if (a.kind == FRT_STFT_PSD) {
                    if (!NOSTORE) row[M / 2] = pm;
                } else {
                    finish_store(row + M / 2, prow + M / 2, M / 2, pm, wg_nyq);
                }
            }
        }
    }
}

template <typename T, int LOG2M, bool DMA>
__global__ void __launch_bounds__(BigPlan<LOG2M>::BLOCK, sizeof(T) == 4 ? 2 : 1) stft_big_kernel(const StftArgs a) {
    using B = BigPlan<LOG2M>;
    using C = cpx<T>;
    static_assert(!DMA || sizeof(T) == 4, "LDS staging is a float32 feature");
    __shared__ C lds[16 * B::RS];
    __shared__ uint32_t lut_lds[256];
    __shared__ __attribute__((aligned(16))) C stage[DMA ? B::M : 1];
    // float64 re-reads its sub-transform twiddles at every use (no registers to hold them): from a copy of the table
    // in LDS — a global load there is an exposed L2 round trip per pass, and it queues behind the row stores
    __shared__ __attribute__((aligned(16))) C tws_lds[sizeof(T) == 8 ? B::MS : 1];
    stft_big_body<T, LOG2M, DMA>(a, lds, lut_lds, stage, tws_lds);
}

}  // namespace frt
