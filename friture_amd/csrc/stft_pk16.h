// stft_pk16.h — K1 for N = 16384, float32, hop N/2 or N/4: stft_pk_kernel (stft_pk.h) with the sixteen 512-point
// sub-transforms factored 16 x 16 x 2 instead of 8 x 8 x 8.  Included by stft.hip after stft_pk.h, whose packed arithmetic,
// LDS sample ring, first stage and row stores it keeps.
//
// stft_pk_kernel's sub-transform phase is bound by the LDS instruction rates (profiles/r04_stft16384_intervals.txt: 48 reads
// and 48 writes per thread and frame in three radix-8 passes at 91 % of the measured ds_read_b64 / ds_write_b64 rates).  Here a
// wavefront still owns two regions (sub-transforms), but each HALF of it takes one: 32 lanes x 16 points,
//     n1 = p + 32 q,  k1 = r + 16 (w + 16 x),  p = u + 2 v:
//     pass 1   X_p[r]    = sum_q y[p + 32 q] W16^(q r),  times W512^(p r)
//     pass 2   T_u[r][w] = sum_v X_(u+2v)[r] W16^(v w),  times W32^(u w)
//     Z[r + 16 w + 256 x] = T_0[r][w] + (-1)^x T_1[r][w]                      — formed by the unpack, which wants Z[k] and
// Z[k + 4096] of one bin anyway: a thread owns the bins k = 4 t + c, k + 4096 and their mirrors M - k, 4096 - k.
// 32 reads and 32 writes per thread and frame in that phase instead of 48 and 48; the unpack still reads 16 values.
// LDS slots of a region (546 per region: the stride-17 exchange needs 542, and 546 mod 32 = 2 spreads the unpack's four
// regions per lane group over the banks):
//     transpose (first stage) and the gather of pass 1:  n1 itself
//     between the passes:  v + 17 r + 272 u   — a write's 16 lanes (v) are contiguous, a read's 32 lanes (r, u) hit
//                          17 r + 16 u mod 32, all different; no address arithmetic beyond an immediate on either side
//     after pass 2:  r + 16 w + 256 u
#pragma once

namespace frt {

struct Pk16Plan {
    static constexpr int LOG2M = 13, M = 1 << LOG2M, MS = M / 16, BLOCK = MS, NW = MS / 64;
    static constexpr int RS = MS + 34;
    static constexpr int REG_BYTES = 16 * RS * 8;                   // 69 888
    static constexpr int RING_OFF = (REG_BYTES + 1023) / 1024 * 1024;
    static constexpr int LUT_OFF = RING_OFF + M * 8;
    static constexpr int LDS_BYTES = LUT_OFF + 1024;                // 137 216
};

// KIND: 0 PSD, 1 dB / normalised (run-time choice), 3 colour image, 4 colour image without the + 1e-30.  HS: ring slots (of 512
// complex) a hop advances: 8 = hop N/2, 4 = hop N/4.
template <int KIND, int HS>
__global__ void __launch_bounds__(Pk16Plan::BLOCK, 2) stft_pk16_kernel(const StftArgs a) {
    using P = Pk16Plan;
    constexpr int M = P::M, MS = P::MS, RS = P::RS;
    constexpr int PH = 16 / HS;                                     // frames until the ring is back in phase
    constexpr bool IMAGE = KIND >= 3, EPS_FREE = KIND == 4;
#if defined(FRT_PKS_NT_BOTH)          // experiment builds: both hops / neither
    constexpr bool kNtRows = true;
#elif defined(FRT_PKS_NT_NONE)
    constexpr bool kNtRows = false;
#else
    constexpr bool kNtRows = HS == 4;                               // non-temporal row stores at hop N/4 (see the stores)
#endif
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

    const pk2* xs = (const pk2*)((const float*)a.x + chan * a.x_stride);
    const pk2* win = (const pk2*)a.window;
    const pk2* tw = (const pk2*)a.tw;          // exp(-2 pi i n / M)
    const pk2* twn = (const pk2*)a.twn;        // exp(-2 pi i k / N)
    const pk2* tws = (const pk2*)a.tws;        // exp(-2 pi i n / 512)
    const float* wgt = (const float*)(IMAGE ? a.wimage : a.weight);
    const float image_gain = (float)a.image_gain, norm_off = (float)a.norm_off, norm_scale = (float)a.norm_scale;

    // sub-transform roles of this lane: half-wave hw takes region wave + 8 hw; inside it lane l5 = v + 16 u holds p = u + 2 v in
    // pass 1 and r = v's value, u in pass 2
    const int hw = lane >> 5, l5 = lane & 31, lv = l5 & 15, lu = l5 >> 4, p = lu + 2 * lv;

    // ---- per-thread constants of a run, in registers ------------------------------------------------------------------
    pk2 winr[16], tw1[15], tw2[15], twur[4];
    float wgr[KIND == 0 ? 1 : 16];
#pragma unroll
    for (int j = 0; j < 16; ++j) winr[j] = win[t + j * MS];
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) tw1[k0 - 1] = tw[(t * k0) & (M - 1)];
#pragma unroll
    for (int r = 1; r < 16; ++r) tw2[r - 1] = tws[(p * r) & (MS - 1)];
    // W32^w, w = 1..15 (the lanes with u = 1 multiply by them): wave-uniform
    pk2 tw3[15];
#pragma unroll
    for (int w = 1; w < 16; ++w) tw3[w - 1] = tws[16 * w];
    // unpack: thread t owns the bins k = 4 t + c (c < 4), k + 4096 and the mirrors M - k, 4096 - k: four 16-byte stores
#pragma unroll
    for (int c = 0; c < 4; ++c) twur[c] = twn[4 * t + c];
    // the twiddle of the self-mirrored pair (M/4, 3M/4), thread 0's: read HERE.  Read inside the frame loop (round 4 .. 6), its use was a
    // vmcnt(0) in wave 0 behind the frame's four row stores — a wait for their write acknowledgements (and the next frame's sample copy)
    // that the other seven wavefronts then spent at barrier A, every frame
    const pk2 tw_quarter = twn[M / 4];
    float wg_q[2] = {0.f, 0.f};                                     // thread 0: bins 2048 and 6144
    if constexpr (KIND != 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * t + c;
            wgr[c] = wgt ? wgt[k] : 0.f;
            wgr[4 + c] = wgt ? wgt[M - k] : 0.f;
            wgr[8 + c] = wgt ? wgt[k + M / 2] : 0.f;
            wgr[12 + c] = wgt ? wgt[M / 2 - k] : 0.f;
        }
        wg_q[0] = wgt ? wgt[M / 4] : 0.f;
        wg_q[1] = wgt ? wgt[3 * M / 4] : 0.f;
    }

    // ---- LDS addresses (bytes) -------------------------------------------------------------------------------------------
    // ring: [wave][slot 0..15][lane], one slot = the wave's 64 complex samples z[64 wave + lane + 512 j]
    const uint32_t ring_wave = sm + (uint32_t)P::RING_OFF + (uint32_t)wave * 8192u;
    const uint32_t ring_lane = ring_wave + lane * 8;
    // transpose: region k0, slot t
    const uint32_t tr_lane = sm + t * 8;
    // sub-transforms
    const uint32_t sub = sm + (wave + 8 * hw) * (RS * 8);
    const uint32_t ga = sub + p * 8;                                // pass-1 gather: + 256 q
    const uint32_t xw = sub + (lv + 272 * lu) * 8;                  // exchange, write side (lane = v, u): + 136 r
    const uint32_t xr = sub + (17 * lv + 272 * lu) * 8;             // exchange, read side (lane = r, u): + 8 v
    const uint32_t fw = sub + (lv + 256 * lu) * 8;                  // after pass 2 (lane = r, u): + 128 w
    // unpack: T_u[r][w] of bin k = k2 + 16 (r + 16 w) sits in region k2, slot r + 16 w + 256 u.  k = 4 t + c: region 4 (t & 3) + c,
    // r + 16 w = t >> 2.  The mirror M - k: region (16 - k2) & 15, r' + 16 w' = (512 - ((4 t + c + 15) >> 4)) & 255 (its x is 1;
    // k = 0 lands on slot 0 of region 0 and is replaced).
    const uint32_t ulo = sm + ((4 * (t & 3)) * RS + (t >> 2)) * 8;
    uint32_t uhi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int u = 4 * t + c, sl = (512 - ((u + 15) >> 4)) & 255;
        uhi[c] = sm + (((16 - (u & 15)) & 15) * RS + sl) * 8;
    }

    // ---- sample copies ----------------------------------------------------------------------------------------------------
    // one copy instruction = 1 KB = ring slots (s, s + 1) of this wave: lanes 0-31 fetch z[64 wave .. + 64) + 512 j, lanes 32-63 the same at j + 1
    const uint32_t copy_lane = (uint32_t)(lane & 31) * 16u + (uint32_t)(lane >> 5) * 4096u;
    auto copy_slots = [&](long long frame, int j_first, int n_slots, int ring_slot_first) {
        const char* src = (const char*)(xs + (frame * a.hop >> 1) + 64 * wave + (long long)j_first * MS);
#pragma unroll
        for (int i = 0; i < n_slots / 2; ++i) {
            const uint32_t dst = ring_wave + (uint32_t)(((ring_slot_first + 2 * i) & 15) * 512);      // LDS address
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :
                         : "v"(copy_lane), "s"(src + (long long)i * 2 * MS * 8), "s"(dst)
                         : "memory", "m0");
        }
    };
    if (nfr > 0) copy_slots(f0, 0, 16, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if FRT_PK_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const unsigned long long run_t0 = __builtin_amdgcn_s_memtime(), run_r0 = __builtin_amdgcn_s_memrealtime();
#define PK_TICK(i)                                                         \
    do {                                                                   \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();     \
        if ((i) >= 0) tacc[(i) < 0 ? 0 : (i)] += now__ - tprev;            \
        tprev = now__;                                                     \
    } while (0)
#else
#define PK_TICK(i) do { } while (0)
#endif

    // ---- first stage of frame g (ph = g mod PH, compile time): samples from the wave's ring slots, window, 16-point DFT over j
    pk2 v[16];
    auto first_stage = [&](auto phc, int g) {
        constexpr int ph = decltype(phc)::value;
        // the copy of this frame's new samples has landed once at most the 4 row stores issued after it (thread 0's two extra
        // bins make it wait for a little more than it must) are outstanding — vector-memory operations retire in order
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = lds_rd(ring_lane + ((j + ph * HS) & 15) * 512) * winr[j];
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
        if (g + 1 < nfr) copy_slots(f0 + g + 1, 16 - HS, HS, ph * HS);
        pk_dft16(v);
    };

    auto frame = [&](auto phc, int g) -> bool {
        if (g >= nfr) return false;
        PK_TICK(-1);
        first_stage(phc, g);
        PK_TICK(0);
        PK_TICK(1);
        __syncthreads();                                            // A: the previous frame's unpack has read the regions
        PK_TICK(2);
        lds_wr(tr_lane, v[0]);
#pragma unroll
        for (int k0 = 1; k0 < 15; k0 += 2) {
            pk_cmul2(v[k0], tw1[k0 - 1], v[k0 + 1], tw1[k0]);
            lds_wr(tr_lane + k0 * (RS * 8), v[k0]);
            lds_wr(tr_lane + (k0 + 1) * (RS * 8), v[k0 + 1]);
        }
        v[15] = pk_cmul(v[15], tw1[14]);
        lds_wr(tr_lane + 15 * (RS * 8), v[15]);
        PK_TICK(3);
        __syncthreads();                                            // B
        PK_TICK(4);
        // ---- 2. sixteen 512-point transforms over n1, one per half-wave: LDS traffic of a wave is executed in order and the
        // accesses are volatile, so the exchange between the two passes needs no fence
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_rd(ga + q * 256);
        pk_dft16(v);                                                // v[r] = X_p[r]
#pragma unroll
        for (int r = 1; r < 15; r += 2) {
            pk_cmul2(v[r], tw2[r - 1], v[r + 1], tw2[r]);
            lds_wr(xw + (r - 1) * 136, v[r - 1]);
            lds_wr(xw + r * 136, v[r]);
        }
        v[15] = pk_cmul(v[15], tw2[14]);
        lds_wr(xw + 14 * 136, v[14]);
        lds_wr(xw + 15 * 136, v[15]);
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
        PK_TICK(5);
        __syncthreads();                                            // C
        PK_TICK(6);
        // ---- 3. Z = T_0 +- T_1 and the conjugate-symmetric unpack of the pairs (k, M - k), (k + 4096, 4096 - k), k = 4 t + c ---
        float* row = (float*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);
        uint32_t* prow = (uint32_t*)row;
        // A = Z[k], B = Z[M-k], wk = exp(-2 pi i k / N):  S = A + conj B, tt = wk (A - conj B);
        // 2 X[k] = S + (-i) tt,  2 conj X[M-k] = S - (-i) tt  (the 1/2 rides in the window table)
        auto pair_powers2 = [&](pk2 A0, pk2 B0, pk2 w0, pk2 A1, pk2 B1, pk2 w1, float (&pw)[4]) {
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
        typedef float pk_f4 __attribute__((ext_vector_type(4), aligned(4)));
        typedef uint32_t pk_u4 __attribute__((ext_vector_type(4), aligned(4)));
        // group 0: bins k0 + c and M - k0 - c; group 1: 4096 + k0 + c and 4096 - k0 - c (k0 = 4 t).  pl[g][c], ph[g][c]: their powers
        pk2 t0[4], t1[4], m0[4], m1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t0[c] = lds_rd(ulo + c * (RS * 8));
            t1[c] = lds_rd(ulo + c * (RS * 8) + 2048);
            m0[c] = lds_rd(uhi[c]);
            m1[c] = lds_rd(uhi[c] + 2048);
        }
        if (t == 0) {                                               // k = 0: Z[M] = Z[0] = T0 + T1 and Z[4096] = T0 - T1
            m0[0] = t0[0];
            m1[0] = -t1[0];
        }
        float pl[2][4], phh[2][4];
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            float pw[4];
            // Z[k] = T0 + T1, Z[M - k] = T0' - T1' (x = 1)
            pair_powers2(t0[c] + t1[c], m0[c] - m1[c], twur[c], t0[c + 1] + t1[c + 1], m0[c + 1] - m1[c + 1], twur[c + 1], pw);
            pl[0][c] = pw[0]; phh[0][c] = pw[1]; pl[0][c + 1] = pw[2]; phh[0][c + 1] = pw[3];
            // Z[k + 4096] = T0 - T1, Z[4096 - k] = T0' + T1'; exp(-2 pi i (k + 4096) / N) = -i exp(-2 pi i k / N)
            pair_powers2(t0[c] - t1[c], m0[c] + m1[c], pk_mul_mi(twur[c]), t0[c + 1] - t1[c + 1], m0[c + 1] + m1[c + 1], pk_mul_mi(twur[c + 1]), pw);
            pl[1][c] = pw[0]; phh[1][c] = pw[1]; pl[1][c + 1] = pw[2]; phh[1][c + 1] = pw[3];
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const int klo = 4 * t + (M / 2) * gq, khi = (gq == 0 ? M : M / 2) - 4 * t;      // bins klo + c and khi - c
            const float* plo = pl[gq];
            const float* phi = phh[gq];
            if constexpr (KIND == 0) {
                // hop N/4 (the widgets' default overlap): a frame writes twice the bytes it reads, and the rows leave non-temporally
                // (measured, profiles/r05_stft16384_nt.txt: 0.121-0.123 -> 0.115-0.116 ms, colour 0.148 -> 0.141; at hop N/2 the same
                // hint costs 6-10 %: plain stores there)
                if constexpr (kNtRows) {
                    __builtin_nontemporal_store(pk_f4{plo[0], plo[1], plo[2], plo[3]}, (pk_f4*)(row + klo));
                    __builtin_nontemporal_store(pk_f4{phi[3], phi[2], phi[1], phi[0]}, (pk_f4*)(row + khi - 3));
                } else {
                    *(pk_f4*)(row + klo) = pk_f4{plo[0], plo[1], plo[2], plo[3]};
                    *(pk_f4*)(row + khi - 3) = pk_f4{phi[3], phi[2], phi[1], phi[0]};
                }
            } else if constexpr (IMAGE) {
                float vv[8];
                uint32_t cc[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    vv[c] = index_value(plo[c], wgr[8 * gq + c]);
                    vv[4 + c] = index_value(phi[c], wgr[8 * gq + 4 + c]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) cc[e] = lut_lds[(int)vv[e]];
                float mm = __builtin_amdgcn_fractf(vv[0]);
#pragma unroll
                for (int e = 1; e < 8; ++e) mm = fminf(mm, __builtin_amdgcn_fractf(vv[e]));
                if (__any(mm < a.edge2)) {                          // within 2 thr above an index edge: one float64 comparison decides
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool near_edge = __builtin_amdgcn_fractf(vv[e]) < a.edge2;
                        const int kk = e < 4 ? klo + e : khi - (e - 4);
                        const int n = exact_colour_index(near_edge, e < 4 ? plo[e] : phi[e - 4], kk, (int)vv[e], a);
                        if (near_edge) cc[e] = lut_lds[n];
                    }
                }
                if constexpr (kNtRows) {
                    __builtin_nontemporal_store(pk_u4{cc[0], cc[1], cc[2], cc[3]}, (pk_u4*)(prow + klo));
                    __builtin_nontemporal_store(pk_u4{cc[7], cc[6], cc[5], cc[4]}, (pk_u4*)(prow + khi - 3));
                } else {
                    *(pk_u4*)(prow + klo) = pk_u4{cc[0], cc[1], cc[2], cc[3]};
                    *(pk_u4*)(prow + khi - 3) = pk_u4{cc[7], cc[6], cc[5], cc[4]};
                }
            } else {
                *(pk_f4*)(row + klo) = pk_f4{finish(plo[0], wgr[8 * gq]), finish(plo[1], wgr[8 * gq + 1]), finish(plo[2], wgr[8 * gq + 2]),
                                             finish(plo[3], wgr[8 * gq + 3])};
                *(pk_f4*)(row + khi - 3) = pk_f4{finish(phi[3], wgr[8 * gq + 7]), finish(phi[2], wgr[8 * gq + 6]),
                                                 finish(phi[1], wgr[8 * gq + 5]), finish(phi[0], wgr[8 * gq + 4])};
            }
        }
        if (t == 0) {
            // the pair (2048, 6144) is its own mirror image: Z[2048] = T0 + T1, Z[6144] = T0 - T1 of region 0, r + 16 w = 128
            const pk2 q0 = lds_rd(sm + 128 * 8), q1 = lds_rd(sm + (128 + 256) * 8);
            float pw[4];
            pair_powers2(q0 + q1, q0 - q1, tw_quarter, q0 + q1, q0 - q1, tw_quarter, pw);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kk = e == 0 ? M / 4 : 3 * M / 4;
                const float pm = pw[e];
                if constexpr (KIND == 0) {
                    row[kk] = pm;
                } else if constexpr (IMAGE) {
                    const float vv = index_value(pm, wg_q[e]);
                    int idx = (int)vv;
                    const bool near_edge = __builtin_amdgcn_fractf(vv) < a.edge2;
                    if (near_edge) idx = exact_colour_index(near_edge, pm, kk, idx, a);
                    prow[kk] = lut_lds[idx];
                } else {
                    row[kk] = finish(pm, wg_q[e]);
                }
            }
        }
        PK_TICK(7);
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
#if FRT_PK_TIMING
    // interval i of this wave, summed over the run's frames (tools/exp/pk_timing.py): 0 copy wait + ring reads + window + DFT16,
    // 1 -, 2 barrier A, 3 transpose writes, 4 barrier B, 5 the two radix-16 passes, 6 barrier C, 7 unpack + stores
    if (lane == 0 && nfr > 0) {
        float* row = (float*)a.out + chan * a.out_cstride + f0 * (M + 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) row[wave * 8 + i] = (float)tacc[i] / (float)nfr;
        row[64 + wave] = (float)(__builtin_amdgcn_s_memtime() - run_t0) / (float)(__builtin_amdgcn_s_memrealtime() - run_r0) * 0.1f;
    }
#endif
#undef PK_TICK
}

}  // namespace frt
