// stft_pk16r.h — K1 for N = 16384, float32, hop N/2 or N/4: stft_pk16_kernel (stft_pk16.h) cut down to TWO workgroups per CU.
// Included by stft.hip (-DFRT_EXPERIMENTS builds only) after stft_pk16.h, whose first stage, 16 x 16 x 2 sub-transforms, LDS exchange layout and unpack it keeps.
//
// stft_pk16_kernel holds 137 KB of LDS (a 64 KB sample ring filled by LDS-DMA beside the 70 KB of exchange regions) and 176 / 208
// registers (122 of them the run's window, twiddle and weight factors): one 512-thread workgroup per CU, two waves per SIMD, and
// nothing runs while the older wave of a SIMD waits at one of the frame's three barriers (profiles/r04_stft16384_intervals.txt:
// 2700-3000 of a frame's 8400 cycles).  The smaller sizes' siblings gained 20-30 % from a second workgroup per CU.  Here:
//
//   1. No LDS ring.  Thread t's first-stage inputs are z[t + 512 j], j < 16, and a hop of HS slots later the same thread wants
//      z[t + 512 (j + HS)]: the re-used samples of a frame are the thread's OWN registers.  The sixteen slots are a register ring
//      (slot of frame-relative j: (j + ph HS) mod 16, ph compile-time as in the LDS ring); the hop's new samples are loaded
//      straight into the slots of the oldest ones (coalesced 8-byte loads, 512 bytes per wave) after the sub-transform phase
//      (its registers are free by then).  Every sample is still read from HBM once per run.  LDS: 70 KB.
//   2. The run's constants leave the registers.  The window streams through the cache once per frame (it is multiplied into the
//      samples as it arrives, in the registers the transform works in).  Of the fifteen twiddle factors W^(t k) of the first
//      stage and the fifteen W^(p r) of the sub-transforms' first pass a thread keeps six each — k in {1, 2, 3, 4, 8, 12} — and
//      multiplies by the two factors of k = a + b for the other nine (one more rounding of 6e-8 on those points; 18 packed
//      products more per frame, issued in the phases the LDS write path bounds).  The colour kinds fetch their sixteen weights
//      per frame.  <= 128 registers: four waves per SIMD.
//
// MEASURED AND NOT SHIPPED (round 5, profiles/r05_stft16384_two_workgroups.txt): PSD 0.077-0.083 ms against stft_pk16_kernel's
// 0.072-0.076 (hop N/2, 32 ch x 2^20), colour 0.107 against 0.083.  The ablation builds say why: with every global access removed two
// workgroups per CU still take 0.055 ms (7300 cycles per frame and CU: ~4150 of packed float32 arithmetic + ~3360 of LDS instructions,
// which do not overlap) — the frame is bound by those, not by waiting, and stft_pk16_kernel's 8400 cycles per frame are within 15 % of
// it; what the second workgroup adds back in memory phases (+0.014 ms for the loads, +0.014 for the stores) is more than it hides.
// Only -DFRT_EXPERIMENTS builds compile it (FRT_STFT_PK16R=1 selects it); parity of that build: tests/test_stft_gpu.py -k large_frame.
#pragma once

#ifndef FRT_PKR_FETCH_PSD           // where a frame requests the hop's new samples of the next one: 0 in its first stage (a frame ahead; the
#define FRT_PKR_FETCH_PSD 1        // slots stay occupied through the sub-transforms), 1 after the sub-transform phase, 2 after the unpack;
#endif                             // 3: no register ring — every frame loads its sixteen slots (the re-used ones from L2) after a one-load-
#ifndef FRT_PKR_FETCH_DB            // per-thread touch of the next hop's lines a frame ahead.  Per kind: the dB and colour kinds' unpack
#define FRT_PKR_FETCH_DB 2         // phases are the tight ones (the colour kind with a ring spills it: 78 registers)
#endif
#ifndef FRT_PKR_FETCH_IMAGE
#define FRT_PKR_FETCH_IMAGE 3
#endif

#ifndef FRT_PKR_ABLATE              // timing experiments (wrong output): 1 no window loads, 2 no per-frame unpack-factor loads, 4 no second
#define FRT_PKR_ABLATE 0           // twiddle products, 8 no row stores, 16 no sample loads after the run's first frame
#endif

#ifndef FRT_PKR_STAGGER             // > 0: the second half of the launch's workgroups (the second resident one of a CU, if the dispatcher fills
#define FRT_PKR_STAGGER 0          // CU after CU) starts this many 64-cycle sleeps late: two workgroups in lockstep stall together
#endif

namespace frt {

struct Pk16rPlan {
    static constexpr int LOG2M = 13, M = 1 << LOG2M, MS = M / 16, BLOCK = MS, NW = MS / 64;
    static constexpr int RS = MS + 34;
    static constexpr int REG_BYTES = 16 * RS * 8;                   // 69 888
    static constexpr int LUT_OFF = (REG_BYTES + 1023) / 1024 * 1024;
    static constexpr int LDS_BYTES = LUT_OFF + 1024;                // 71 680: two workgroups per CU
};

// ---- buffer accesses: descriptor in scalar registers, 32-bit lane offset, scalar offset ---------------------------------------
typedef __amdgpu_buffer_rsrc_t pkr_rsrc;
typedef float pk_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t pk_u4 __attribute__((ext_vector_type(4)));
// raw buffer (stride 0), no bounds to speak of (the kernel addresses only what the host sized), gfx950 data format word
__device__ __forceinline__ pkr_rsrc pkr_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ pk2 pkr_load8(pkr_rsrc r, uint32_t lane_off, uint32_t soff) {
    return __builtin_bit_cast(pk2, __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, soff, 0));
}
__device__ __forceinline__ pk_f4 pkr_load16(pkr_rsrc r, uint32_t lane_off, uint32_t soff) {
    return __builtin_bit_cast(pk_f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, soff, 0));
}
template <bool NT, typename V>
__device__ __forceinline__ void pkr_store16(pkr_rsrc r, uint32_t lane_off, uint32_t soff, V v) {
    const pk_u4 bits = __builtin_bit_cast(pk_u4, v);
    __builtin_amdgcn_raw_buffer_store_b128(bits, r, lane_off, soff, NT ? 2 : 0);      // aux 2: nt
    // The data registers stay untouched for two more issue slots.  The compiler pads the "store of more than 8 bytes, then a write of
    // its data registers" hazard only when the scalar offset is an immediate; with a register there it emitted v_mov_b32 v0 right
    // behind buffer_store_dwordx4 v[0:3] and, with the chip full, the first word of such stores arrived overwritten (session r5r:
    // 896 of 33 M bins, only in the launch's second round of workgroups, never twice the same).
    asm volatile("s_nop 1" :: "v"(bits));
}

// v[k] *= W^k for k = 1 .. 15 from the six kept powers: b[0..2] = W^1, W^2, W^3 and a4[0..2] = W^4, W^8, W^12
__device__ __forceinline__ void pkr_twiddle16(pk2 (&v)[16], const pk2 (&b)[3], const pk2 (&a4)[3]) {
    pk_cmul2(v[1], b[0], v[2], b[1]);
    pk_cmul2(v[3], b[2], v[4], a4[0]);
    pk_cmul2(v[8], a4[1], v[12], a4[2]);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        pk_cmul2(v[4 * q + 1], b[0], v[4 * q + 2], b[1]);
        v[4 * q + 3] = pk_cmul(v[4 * q + 3], b[2]);
        if constexpr (!(FRT_PKR_ABLATE & 4)) {
            pk_cmul2(v[4 * q + 1], a4[q - 1], v[4 * q + 2], a4[q - 1]);
            v[4 * q + 3] = pk_cmul(v[4 * q + 3], a4[q - 1]);
        }
    }
}

// KIND: 0 PSD, 1 dB / normalised (run-time choice), 3 colour image, 4 colour image without the + 1e-30.  HS: slots (of 512 complex)
// a hop advances: 8 = hop N/2, 4 = hop N/4.
template <int KIND, int HS>
__global__ void __launch_bounds__(Pk16rPlan::BLOCK, 4) stft_pk16r_kernel(const StftArgs a) {
    using P = Pk16rPlan;
    constexpr int M = P::M, MS = P::MS, RS = P::RS;
    constexpr int PH = 16 / HS;                                     // frames until the register ring is back in phase
    constexpr bool IMAGE = KIND >= 3, EPS_FREE = KIND == 4;
    constexpr bool kNtRows = HS == 4;                               // non-temporal row stores at hop N/4 (stft_pk16.h)
    constexpr int kFetch = KIND == 0 ? FRT_PKR_FETCH_PSD : IMAGE ? FRT_PKR_FETCH_IMAGE : FRT_PKR_FETCH_DB;
    constexpr bool kRing = kFetch != 3;
    __shared__ __attribute__((aligned(1024))) char smem[P::LDS_BYTES];
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the block
    uint32_t* const lut_lds = (uint32_t*)(smem + P::LUT_OFF);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if constexpr (IMAGE) {
        if (t < 256) lut_lds[t] = a.lut[t];                         // visible after the first frame's barriers
    }

    const int gg = blockIdx.x;
    const int chan = gg / a.runs_per_channel;
    const int run = gg - chan * a.runs_per_channel;
    const long long f0 = a.frame_base + (long long)run * a.run;
    int nfr = (int)(a.n_frames - f0 < (long long)a.run ? a.n_frames - f0 : (long long)a.run);
    if (gg >= a.n_groups || nfr < 0) nfr = 0;

    // Global accesses are buffer operations: a wave-uniform descriptor (four scalar registers per array), a scalar offset for the
    // slot / frame and ONE 32-bit lane offset — o8 for the arrays of complex values indexed by t, o16 / r16 for the ascending /
    // descending groups of four bins.  (As flat accesses every slot costs a 64-bit address pair: 60 registers spilled.)
    const uint32_t o8 = (uint32_t)t * 8u, o16 = (uint32_t)t * 16u, r16 = (uint32_t)(MS - 1 - t) * 16u;
    const long long hop2 = a.hop >> 1;                              // complex samples per hop
    const pkr_rsrc xs_rs = pkr_make_rsrc((const pk2*)((const float*)a.x + chan * a.x_stride) + f0 * hop2);   // z[...] of the run's first frame
    const pkr_rsrc win_rs = pkr_make_rsrc(a.window);
    const pkr_rsrc out_rs = pkr_make_rsrc((const float*)a.out + chan * a.out_cstride + f0 * (M + 1));       // the run's first row
    const pk2* tw = (const pk2*)a.tw;          // exp(-2 pi i n / M)
    const pk2* twn = (const pk2*)a.twn;        // exp(-2 pi i k / N)
    const pk2* tws = (const pk2*)a.tws;        // exp(-2 pi i n / 512)
    const float* wgt = (const float*)(IMAGE ? a.wimage : a.weight);
    const pkr_rsrc wgt_rs = pkr_make_rsrc(wgt);
    const float image_gain = (float)a.image_gain, norm_off = (float)a.norm_off, norm_scale = (float)a.norm_scale;

    // sub-transform roles of this lane (stft_pk16.h): half-wave hw takes region wave + 8 hw; inside it lane l5 = v + 16 u holds
    // p = u + 2 v in pass 1 and r = v's value, u in pass 2
    const int hw = lane >> 5, l5 = lane & 31, lv = l5 & 15, lu = l5 >> 4, p = lu + 2 * lv;

    // ---- per-thread constants of a run ------------------------------------------------------------------------------------
    pk2 t1b[3], t1a[3], t2b[3], t2a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        t1b[k] = tw[(t * (k + 1)) & (M - 1)];
        t1a[k] = tw[(t * 4 * (k + 1)) & (M - 1)];
        t2b[k] = tws[(p * (k + 1)) & (MS - 1)];
        t2a[k] = tws[(p * 4 * (k + 1)) & (MS - 1)];
    }
    // W32^w, w = 1..15 (the lanes with u = 1 multiply by them): wave-uniform
    pk2 tw3[15];
#pragma unroll
    for (int w = 1; w < 16; ++w) tw3[w - 1] = tws[16 * w];
    // unpack: thread t owns the bins k = 4 t + c (c < 4), k + 4096 and the mirrors M - k, 4096 - k: four 16-byte stores; their
    // factors exp(-2 pi i k / N) arrive per frame (two 16-byte loads)
    const pkr_rsrc twn_rs = pkr_make_rsrc(twn);

    // ---- LDS addresses (bytes), as in stft_pk16_kernel --------------------------------------------------------------------
    const uint32_t tr_lane = sm + t * 8;                            // transpose: region k0, slot t
    const uint32_t sub = sm + (wave + 8 * hw) * (RS * 8);
    const uint32_t ga = sub + p * 8;                                // pass-1 gather: + 256 q
    const uint32_t xw = sub + (lv + 272 * lu) * 8;                  // exchange, write side (lane = v, u): + 136 r
    const uint32_t xr = sub + (17 * lv + 272 * lu) * 8;             // exchange, read side (lane = r, u): + 8 v
    const uint32_t fw = sub + (lv + 256 * lu) * 8;                  // after pass 2 (lane = r, u): + 128 w
    const uint32_t ulo = sm + ((4 * (t & 3)) * RS + (t >> 2)) * 8;
    // the mirrors M - k, k = 4 t + c: region (16 - (k & 15)) & 15, slot (512 - ((k + 15) >> 4)) & 255 (stft_pk16.h).  For c = 1, 2, 3 the
    // slot is the same and the region falls by one per c (16 - 4 (t & 3) - c stays inside 1 .. 15): one lane base and immediates
    uint32_t uhi0, uhi3;
    {
        const int sl0 = (512 - ((4 * t + 15) >> 4)) & 255, sl3 = (512 - ((4 * t + 18) >> 4)) & 255;
        uhi0 = sm + (((16 - ((4 * t) & 15)) & 15) * RS + sl0) * 8;
        uhi3 = sm + ((13 - ((4 * t) & 15)) * RS + sl3) * 8;
    }
    auto uhi = [&](int c) -> uint32_t { return c == 0 ? uhi0 : uhi3 + (uint32_t)((3 - c) * (RS * 8)); };

    // ---- the register ring of this thread's samples: slot j of the run's first frame = z[t + 512 j] -----------------------
    pk2 s[kRing ? 16 : 1];
    const uint32_t hop_bytes = (uint32_t)hop2 * 8u;
    if constexpr (kRing) {
        if (nfr > 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) s[j] = pkr_load8(xs_rs, o8, (uint32_t)(j * (MS * 8)));
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) s[j] = pk2{0.f, 0.f};
        }
    }
    // (no ring) one 4-byte load per thread in every 64 bytes of the next frame's new samples: they are in L2 when that frame asks
    uint32_t touched = 0u;
    const uint32_t touch_lane = ((uint32_t)t * 64u) & (uint32_t)(HS * MS * 8 - 1);
    // the hop's new samples of frame g + 1 into the slots frame g (phase ph) read its oldest samples from
    auto fetch_next = [&](auto phc, int g) {
        constexpr int ph = decltype(phc)::value;
        if constexpr (kRing) {
            if (g + 1 < nfr && !(FRT_PKR_ABLATE & 16)) {
                const uint32_t src = (uint32_t)(g + 1) * hop_bytes + (uint32_t)((16 - HS) * MS * 8);
#pragma unroll
                for (int i = 0; i < HS; ++i) s[(ph * HS + i) & 15] = pkr_load8(xs_rs, o8, src + (uint32_t)(i * (MS * 8)));
            }
        }
    };

    if constexpr (FRT_PKR_STAGGER > 0) {
        if (2 * gg >= a.n_groups) {
            constexpr int kFull = FRT_PKR_STAGGER / 127, kRest = FRT_PKR_STAGGER % 127;
            for (int i = 0; i < kFull; ++i) __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(kRest);
        }
    }
    pk2 v[16];
    auto frame = [&](auto phc, int g) -> bool {
        constexpr int ph = decltype(phc)::value;
        if (g >= nfr) return false;
        // ---- 1. window (streamed), 16-point DFT over j ---------------------------------------------------------------------
        if constexpr (kRing) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (FRT_PKR_ABLATE & 1) ? pk2{0.5f, 0.5f} : pkr_load8(win_rs, o8, (uint32_t)(j * (MS * 8)));
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = v[j] * s[(j + ph * HS) & 15];
        } else {
            pk2 x[16];
            const uint32_t src = (uint32_t)g * hop_bytes;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                x[j] = (FRT_PKR_ABLATE & 16) && (j < 16 - HS || g > 0) ? pk2{0.25f, -0.25f} : pkr_load8(xs_rs, o8, src + (uint32_t)(j * (MS * 8)));
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (FRT_PKR_ABLATE & 1) ? pk2{0.5f, 0.5f} : pkr_load8(win_rs, o8, (uint32_t)(j * (MS * 8)));
            asm volatile("" :: "v"(touched));                       // (the touch of a frame ago is waited for here, behind this frame's loads)
            if (g + 1 < nfr)
                touched = __builtin_amdgcn_raw_buffer_load_b32(xs_rs, touch_lane, src + hop_bytes + (uint32_t)((16 - HS) * MS * 8), 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = v[j] * x[j];
        }
        if constexpr (kFetch == 0) fetch_next(phc, g);
        pk_dft16(v);
        __syncthreads();                                            // A: the previous frame's unpack has read the regions
        pkr_twiddle16(v, t1b, t1a);
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) lds_wr(tr_lane + k0 * (RS * 8), v[k0]);
        __syncthreads();                                            // B
        // ---- 2. sixteen 512-point transforms over n1, one per half-wave --------------------------------------------------
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_rd(ga + q * 256);
        pk_dft16(v);                                                // v[r] = X_p[r]
        pkr_twiddle16(v, t2b, t2a);
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_wr(xw + r * 136, v[r]);
#pragma unroll
        for (int vv = 0; vv < 16; ++vv) v[vv] = lds_rd(xr + vv * 8);
        pk_dft16(v);                                                // v[w] = T_u[r][w]
        if (lu) {
#pragma unroll
            for (int w = 1; w < 15; w += 2) pk_cmul2_s(v[w], tw3[w - 1], v[w + 1], tw3[w]);
            v[15] = pk_cmul_s(v[15], tw3[14]);
        }
#pragma unroll
        for (int w = 0; w < 16; ++w) lds_wr(fw + w * 128, v[w]);
        if constexpr (kFetch == 1) fetch_next(phc, g);
        __syncthreads();                                            // C
        // ---- 3. Z = T_0 +- T_1 and the conjugate-symmetric unpack of the pairs (k, M - k), (k + 4096, 4096 - k), k = 4 t + c ---
        // Two bins c, c + 1 at a time (their LDS values, unpack factors and weights are fetched, used and dropped before the next
        // two: the phase's registers are what keeps the colour kinds at four waves per SIMD); the sixteen results wait in outv
        // for the four 16-byte stores.  outv[gq][0][c]: bin klo + c, outv[gq][1][c]: bin khi - c; klo = 4 t + 4096 gq,
        // khi = (M or M/2) - 4 t.
        float* row = (float*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);      // (thread 0's two single bins)
        uint32_t* prow = (uint32_t*)row;
        const uint32_t row_off = (uint32_t)g * (uint32_t)((M + 1) * 4);
        auto pair_powers2 = [&](pk2 A0, pk2 B0, pk2 w0, pk2 A1, pk2 B1, pk2 w1, float (&pw)[4]) {
            // A = Z[k], B = Z[M-k], wk = exp(-2 pi i k / N):  S = A + conj B, tt = wk (A - conj B);
            // 2 X[k] = S + (-i) tt,  2 conj X[M-k] = S - (-i) tt  (the 1/2 rides in the window table)
            const pk2 S0 = pk_add_conj(A0, B0), S1 = pk_add_conj(A1, B1);
            pk2 t0 = pk_sub_conj(A0, B0), t1 = pk_sub_conj(A1, B1);
            pk_cmul2(t0, w0, t1, w1);
            const pk2 xk0 = pk_add_mi(S0, t0), xm0 = pk_sub_mi(S0, t0), xk1 = pk_add_mi(S1, t1), xm1 = pk_sub_mi(S1, t1);
            const pk2 k0 = xk0 * xk0, m0 = xm0 * xm0, k1 = xk1 * xk1, m1 = xm1 * xm1;
            pw[0] = k0.x + k0.y;
            pw[1] = m0.x + m0.y;
            pw[2] = k1.x + k1.y;
            pw[3] = m1.x + m1.y;
        };
        auto finish = [&](float pp, float w) -> float {            // dB kinds
            float vv = db10<float>(pp) + w;
            if (a.kind == FRT_STFT_NORM) vv = (vv + norm_off) * norm_scale;
            return vv;
        };
        auto index_value = [&](float pp, float w) -> float {
            return clamp_index(image_gain * log2_t(EPS_FREE ? pp : pp + 1e-30f) + w);
        };
        typedef typename std::conditional<IMAGE, uint32_t, float>::type out_t;
        out_t outv[2][2][4];
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            // exp(-2 pi i (4 t + c) / N) and the next one: 16 bytes
            const pk_f4 uu = (FRT_PKR_ABLATE & 2) ? pk_f4{1.f, 0.f, 0.f, 1.f} : pkr_load16(twn_rs, o16 * 2u, (uint32_t)(c * 8));
            const pk2 twu0 = {uu.x, uu.y}, twu1 = {uu.z, uu.w};
            // weights of the bins klo + c, klo + c + 1 (wl[gq]) and khi - c - 1, khi - c (wh[gq]: descending addresses)
            pk2 wl[2], wh[2];
            if constexpr (KIND != 0) {
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    wl[gq] = wh[gq] = pk2{0.f, 0.f};
                    if (wgt) {
                        wl[gq] = pkr_load8(wgt_rs, o16, (uint32_t)(((M / 2) * gq + c) * 4));
                        wh[gq] = pkr_load8(wgt_rs, r16, (uint32_t)(((gq == 0 ? M : M / 2) - c - 1 - 4 * (MS - 1)) * 4));
                    }
                }
            }
            pk2 t0[2], t1[2], m0[2], m1[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                t0[e] = lds_rd(ulo + (c + e) * (RS * 8));
                t1[e] = lds_rd(ulo + (c + e) * (RS * 8) + 2048);
                m0[e] = lds_rd(uhi(c + e));
                m1[e] = lds_rd(uhi(c + e) + 2048);
            }
            if (c == 0 && t == 0) {                                 // k = 0: Z[M] = Z[0] = T0 + T1 and Z[4096] = T0 - T1
                m0[0] = t0[0];
                m1[0] = -t1[0];
            }
            // pw[gq]: powers of the bins klo + c, khi - c, klo + c + 1, khi - c - 1
            float pw[2][4];
            // Z[k] = T0 + T1, Z[M - k] = T0' - T1' (x = 1)
            pair_powers2(t0[0] + t1[0], m0[0] - m1[0], twu0, t0[1] + t1[1], m0[1] - m1[1], twu1, pw[0]);
            // Z[k + 4096] = T0 - T1, Z[4096 - k] = T0' + T1'; exp(-2 pi i (k + 4096) / N) = -i exp(-2 pi i k / N)
            pair_powers2(t0[0] - t1[0], m0[0] + m1[0], pk_mul_mi(twu0), t0[1] - t1[1], m0[1] + m1[1], pk_mul_mi(twu1), pw[1]);
            if constexpr (KIND == 0) {
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    outv[gq][0][c] = pw[gq][0]; outv[gq][1][c] = pw[gq][1];
                    outv[gq][0][c + 1] = pw[gq][2]; outv[gq][1][c + 1] = pw[gq][3];
                }
            } else if constexpr (!IMAGE) {
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    outv[gq][0][c] = finish(pw[gq][0], wl[gq].x); outv[gq][1][c] = finish(pw[gq][1], wh[gq].y);
                    outv[gq][0][c + 1] = finish(pw[gq][2], wl[gq].y); outv[gq][1][c + 1] = finish(pw[gq][3], wh[gq].x);
                }
            } else {
                // four bins at a time (i as in pw): index value, LUT entry; within 2 thr above an index edge one float64 comparison
                // decides (per group of four: what is live across that rare path are four powers and index values, not sixteen)
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    float vv[4];
                    uint32_t cc[4];
                    vv[0] = index_value(pw[gq][0], wl[gq].x);
                    vv[1] = index_value(pw[gq][1], wh[gq].y);
                    vv[2] = index_value(pw[gq][2], wl[gq].y);
                    vv[3] = index_value(pw[gq][3], wh[gq].x);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cc[i] = lut_lds[(int)vv[i]];
                    const float mm = fminf(fminf(__builtin_amdgcn_fractf(vv[0]), __builtin_amdgcn_fractf(vv[1])),
                                           fminf(__builtin_amdgcn_fractf(vv[2]), __builtin_amdgcn_fractf(vv[3])));
                    if (__any(mm < a.edge2)) {
                        const int klo = 4 * t + (M / 2) * gq, khi = (gq == 0 ? M : M / 2) - 4 * t;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int kk = (i & 1) ? khi - c - (i >> 1) : klo + c + (i >> 1);
                            const bool near_edge = __builtin_amdgcn_fractf(vv[i]) < a.edge2;
                            const int n = exact_colour_index(near_edge, pw[gq][i], kk, (int)vv[i], a);
                            if (near_edge) cc[i] = lut_lds[n];
                        }
                    }
                    outv[gq][0][c] = cc[0]; outv[gq][1][c] = cc[1];
                    outv[gq][0][c + 1] = cc[2]; outv[gq][1][c + 1] = cc[3];
                }
            }
        }
        typedef out_t __attribute__((ext_vector_type(4))) out4;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            // bins klo + c at row + klo0 + 4 t; bins khi - 3 + c at row + khi0 - 3 - 4 t = (row + khi0 - 3 - 4 (MS - 1)) + 4 (MS - 1 - t)
            constexpr bool nt = kNtRows && (KIND == 0 || IMAGE);
            if constexpr (FRT_PKR_ABLATE & 8) {
                if (outv[gq][0][0] != (out_t)12345) continue;      // (never true in practice: the stores stay in the code, not in the run)
            }
            pkr_store16<nt>(out_rs, o16, row_off + (uint32_t)((M / 2) * gq * 4), out4{outv[gq][0][0], outv[gq][0][1], outv[gq][0][2], outv[gq][0][3]});
            pkr_store16<nt>(out_rs, r16, row_off + (uint32_t)(((gq == 0 ? M : M / 2) - 3 - 4 * (MS - 1)) * 4),
                            out4{outv[gq][1][3], outv[gq][1][2], outv[gq][1][1], outv[gq][1][0]});
        }
        if (t == 0) {
            // the pair (2048, 6144) is its own mirror image: Z[2048] = T0 + T1, Z[6144] = T0 - T1 of region 0, r + 16 w = 128
            const pk2 q0 = lds_rd(sm + 128 * 8), q1 = lds_rd(sm + (128 + 256) * 8);
            float pw[4];
            pair_powers2(q0 + q1, q0 - q1, twn[M / 4], q0 + q1, q0 - q1, twn[M / 4], pw);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kk = e == 0 ? M / 4 : 3 * M / 4;
                const float pm = pw[e];
                const float wq = (KIND != 0 && wgt) ? wgt[kk] : 0.f;
                if constexpr (KIND == 0) {
                    row[kk] = pm;
                } else if constexpr (IMAGE) {
                    const float vv = index_value(pm, wq);
                    int idx = (int)vv;
                    const bool near_edge = __builtin_amdgcn_fractf(vv) < a.edge2;
                    if (near_edge) idx = exact_colour_index(near_edge, pm, kk, idx, a);
                    prow[kk] = lut_lds[idx];
                } else {
                    row[kk] = finish(pm, wq);
                }
            }
        }
        if constexpr (kFetch == 2) fetch_next(phc, g);
        return true;
    };
    for (int g = 0; g < nfr; g += PH) {
        if (!frame(std::integral_constant<int, 0>{}, g)) break;
        if (!frame(std::integral_constant<int, 1>{}, g + 1)) break;
        if constexpr (PH > 2) {
            if (!frame(std::integral_constant<int, 2>{}, g + 2)) break;
            if (!frame(std::integral_constant<int, 3>{}, g + 3)) break;
        }
    }
}

}  // namespace frt
