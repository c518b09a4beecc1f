// stft_pk16s.h — K1 for N = 8192 / 4096 / 2048, float32, hop N/2 or N/4, rows on 16-byte boundaries: the structure of
// stft_pk16_kernel (stft_pk16.h) one, two and three sizes down, ONE template over the size.  M = N/2 = 16 Ms: Ms threads, thread t takes
// the 16-point DFT over z[t + Ms j] (LDS sample ring, packed arithmetic: stft_pk.h), transposes through LDS; the sixteen Ms-point
// transforms over t are 16 x L inside L lanes (L = Ms / 16 = 16 / 8 / 4), one exchange between their two passes:
//     n1 = p + L q,  k1 = r + 16 w:   X_p[r] = sum_q y[p + L q] W16^(q r), times W_Ms^(p r)      (lane p, 16 points)
//                                     Z[r + 16 w] = sum_p X_p[r] W_L^(p w)                       (L = 16: lane r, 16 points;
//                                                                                                 L = 8: lane l, r = l and l + 8;
//                                                                                                 L = 4: lane l, r = l + 4 h, h < 4)
// and the unpack is stft_pk_kernel's (bins k = 4 t + c + (M/4) g and their mirrors M - k, four 16-byte stores per thread).
//   N = 8192: 71 KB of LDS, 4 wavefronts per workgroup — TWO workgroups per CU, which drift apart: the LDS-bound passes of one run beside
//             the vector-bound first stage and unpack of the other, what the three barriers per frame forbid inside one workgroup;
//   N = 4096: 38 KB, two wavefronts — FOUR workgroups per CU;   N = 2048: 22 KB, a workgroup is a wavefront (its barriers cost
//             nothing) — seven or eight per CU.
// LDS slots of a region: n1 for the transpose and the first gather, r + 16 w after the second pass, and between the passes
//   L = 16 (290 slots per region): v + 17 r.  A lane group of 32 covers two regions: quarter-waves 0, 1 take regions R and R + 8, whose
//          bases differ by 16 banks (290 mod 32 = 2), so the two quarters' 16 slots tile the 32 banks in every access; the unpack's four
//          regions per lane group sit 8 a + 2 c banks apart;
//   L = 8 (region `reg` starts at slot 160 reg + 8 ((reg + (reg >> 2)) & 3): four consecutive regions — a lane group of 32 in the passes —
//          and the four regions 4 a + c of the unpack both start 0, 8, 16, 24 banks apart): (p ^ (r & 7)) + 8 r — a write's eight lanes of
//          a region are a contiguous run, a read's eight lanes land two per run of eight with different low bits in every region;
//   L = 4 (region `reg` starts at slot 96 reg + 4 g(reg), g = reg's low three bits with bit 1 flipped by bit 3: eight consecutive regions
//          start 0, 4, .., 28 banks apart in some order, four consecutive regions 0, 4, 8, 12 mod 16, the unpack's four regions 4 a + c
//          0, 8, 16, 24 apart): (p ^ (r & 3)) + 4 r.
// tests/test_lds_layouts.py asserts these maps under gfx950's bank rules.
#pragma once

namespace frt {

template <int LOG2M_>
struct Pk16sPlan {
    static_assert(LOG2M_ >= 10 && LOG2M_ <= 12, "N = 2048, 4096, 8192");
    static constexpr int LOG2M = LOG2M_, M = 1 << LOG2M, MS = M / 16, BLOCK = MS, NW = MS / 64;
    static constexpr int L = MS / 16;                               // lanes of a sub-transform
    static constexpr int RS = L == 16 ? MS + 34 : MS + 32;          // L < 16: a multiple of 32, the bank offsets are explicit (regbase)
    static constexpr int REG_BYTES = 16 * RS * 8 + (L == 16 ? 0 : 256);      // 37 120 / 20 736 / 12 544
    static constexpr int RING_OFF = (REG_BYTES + 1023) / 1024 * 1024;
    static constexpr int LUT_OFF = RING_OFF + M * 8;
    static constexpr int LDS_BYTES = LUT_OFF + 1024;                // 71 680 / 38 912 / 22 528
    // first slot of a region (see the header comment)
    __host__ __device__ static constexpr int regbase(int reg) {
        return L == 16 ? RS * reg : L == 8 ? RS * reg + 8 * ((reg + (reg >> 2)) & 3) : RS * reg + 4 * ((reg & 5) | ((((reg >> 1) ^ (reg >> 3)) & 1) << 1));
    }
};

template <int LOG2M, int KIND, int HS>
__global__ void __launch_bounds__(Pk16sPlan<LOG2M>::BLOCK, 2) stft_pk16s_kernel(const StftArgs a) {
    using P = Pk16sPlan<LOG2M>;
    constexpr int M = P::M, MS = P::MS, RS = P::RS, L = P::L;
    constexpr int PH = 16 / HS;                                     // frames until the ring is back in phase
    constexpr bool IMAGE = KIND >= 3, EPS_FREE = KIND == 4;
#if defined(FRT_PKS_NT_BOTH)
    constexpr bool kNtRows = true;
#elif defined(FRT_PKS_NT_NONE)
    constexpr bool kNtRows = false;
#else
    constexpr bool kNtRows = HS == 4;                               // non-temporal row stores at hop N/4 (stft_pk16.h; profiles/r05_stft16384_nt.txt)
#endif
    __shared__ __attribute__((aligned(1024))) char smem[P::LDS_BYTES];
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the block
    uint32_t* const lut_lds = (uint32_t*)(smem + P::LUT_OFF);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if constexpr (IMAGE) {                                          // MS threads, 256 entries; visible after the first frame's barriers
#pragma unroll
        for (int i = 0; i < 256 / MS; ++i) lut_lds[t + MS * i] = a.lut[t + MS * i];
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
    const pk2* tws = (const pk2*)a.tws;        // exp(-2 pi i n / MS)
    const float* wgt = (const float*)(IMAGE ? a.wimage : a.weight);
    const float image_gain = (float)a.image_gain, norm_off = (float)a.norm_off, norm_scale = (float)a.norm_scale;

    // sub-transform roles: lane lp of a group of L lanes holds p in pass 1; in pass 2 r = lp (L = 16), lp and lp + 8 (L = 8),
    // lp + 4 h (L = 4).  The group's region: L = 16: quarter-wave qw takes region wave + 4 (qw >> 1) + 8 (qw & 1); L = 8: the eight lanes
    // 8 g .. 8 g + 7 of wave w take region 8 w + g; L = 4: the four lanes 4 g .. 4 g + 3 take region g
    const int qw = lane >> 4, lp = lane & (L - 1);
    const int region = L == 16 ? wave + 4 * (qw >> 1) + 8 * (qw & 1) : L == 8 ? 8 * wave + (lane >> 3) : lane >> 2;

    // ---- per-thread constants of a run, in registers ------------------------------------------------------------------
    pk2 winr[16], tw1[15], tw2[15], twur[8];
    float wgr[KIND == 0 ? 1 : 16];
#pragma unroll
    for (int j = 0; j < 16; ++j) winr[j] = win[t + j * MS];
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) tw1[k0 - 1] = tw[(t * k0) & (M - 1)];
#pragma unroll
    for (int r = 1; r < 16; ++r) tw2[r - 1] = tws[(lp * r) & (MS - 1)];
    // unpack: thread t owns the bin pairs (k, M - k), k = 4 t + c + (M/4) g (c < 4, g < 2): four consecutive bins per 16-byte store
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) twur[4 * g + c] = twn[4 * t + c + (M / 4) * g];
    float wg_nyq = 0.f;
    if constexpr (KIND != 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 4 * t + (q & 3) + (M / 4) * (q >> 2);
            wgr[q] = wgt ? wgt[k] : 0.f;
            wgr[8 + q] = wgt ? wgt[M - k] : 0.f;
        }
        wg_nyq = wgt ? wgt[M / 2] : 0.f;
    }

    // ---- LDS addresses (bytes) -------------------------------------------------------------------------------------------
    const uint32_t ring_wave = sm + (uint32_t)P::RING_OFF + (uint32_t)wave * (16u * 512u);      // [wave][slot 0..15][lane]
    const uint32_t ring_lane = ring_wave + lane * 8;
    const uint32_t tr_lane = sm + t * 8;                            // transpose: region k0 (+ regbase(k0) * 8), slot t
    const uint32_t sub = sm + (uint32_t)(L == 16 ? region * RS : RS * region + (L == 8 ? 8 * ((region + (region >> 2)) & 3)
                                                                                       : 4 * ((region & 5) | ((((region >> 1) ^ (region >> 3)) & 1) << 1)))) * 8u;
    const uint32_t ga = sub + lp * 8;                               // pass-1 gather: + 8 L q; after pass 2: L = 16 + 128 w, L < 16 + 8 L h + 128 w
    // exchange between the passes.  L = 16: slot v + 17 r — write side (lane = v) ga + 136 r, read side (lane = r) xr + 8 v.
    // L < 16: slot (p ^ c) + L r, c = r & (L - 1) — write xa[c] + 8 L r, read X_p'[lp + L h]: xa[p'] + 8 L lp + 8 L L h
    const uint32_t xr = sub + (17 * lp) * 8;
    uint32_t xa[L == 16 ? 1 : L];
    if constexpr (L < 16) {
#pragma unroll
        for (int c = 0; c < L; ++c) xa[c] = sub + (uint32_t)(lp ^ c) * 8u;
    }
    // unpack: Z[k] = region k & 15, slot k >> 4.  k = 4 t + c + (M/4) g: region 4 (t & 3) + c, slot (t >> 2) + (M/64) g.  Z[M - k]: with
    // u = 4 t + c, region (16 - (u & 15)) & 15 and slot MS - (M/64) g - ((u + 15) >> 4) (u = 0, g = 0: Z[M] = Z[0], replaced; the
    // address read instead is slot MS of region 0, inside the region's padding).  L = 16: the four regions of the low side are
    // immediates off one lane base
    uint32_t ulo[L == 16 ? 1 : 4], uhi[4];
    if constexpr (L == 16) ulo[0] = sm + ((4 * (t & 3)) * RS + (t >> 2)) * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int u = 4 * t + c, sl = MS - ((u + 15) >> 4), rl = 4 * (t & 3) + c, rh = (16 - (u & 15)) & 15;
        if constexpr (L < 16) ulo[c] = sm + (uint32_t)(P::regbase(rl) + (t >> 2)) * 8u;
        uhi[c] = sm + (uint32_t)(P::regbase(rh) + sl) * 8u;
    }

    // ---- sample copies: one copy instruction = 1 KB = ring slots (s, s + 1) of this wave -----------------------------------
    const uint32_t copy_lane = (uint32_t)(lane & 31) * 16u + (uint32_t)(lane >> 5) * (uint32_t)(MS * 8);
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

    pk2 v[16];
    auto first_stage = [&](auto phc, int g) {
        constexpr int ph = decltype(phc)::value;
        // the copy of this frame's new samples has landed once at most the 4 row stores issued after it are outstanding (thread 0's
        // extra bin makes its wave wait for one store more than it must) — vector-memory operations retire in order
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
        first_stage(phc, g);
        __syncthreads();                                            // A: the previous frame's unpack has read the regions
        lds_wr(tr_lane + P::regbase(0) * 8, v[0]);
#pragma unroll
        for (int k0 = 1; k0 < 15; k0 += 2) {
            pk_cmul2(v[k0], tw1[k0 - 1], v[k0 + 1], tw1[k0]);
            lds_wr(tr_lane + P::regbase(k0) * 8, v[k0]);
            lds_wr(tr_lane + P::regbase(k0 + 1) * 8, v[k0 + 1]);
        }
        v[15] = pk_cmul(v[15], tw1[14]);
        lds_wr(tr_lane + P::regbase(15) * 8, v[15]);
        __syncthreads();                                            // B
        // ---- 2. sixteen Ms-point transforms over n1, one per L lanes (LDS traffic of a wave is executed in order, the
        // accesses are volatile: the exchange needs no fence)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = lds_rd(ga + q * (8 * L));
        pk_dft16(v);                                                // v[r] = X_p[r]
        if constexpr (L == 16) {
#pragma unroll
            for (int r = 1; r < 15; r += 2) {
                pk_cmul2(v[r], tw2[r - 1], v[r + 1], tw2[r]);
                lds_wr(ga + (r - 1) * 136, v[r - 1]);
                lds_wr(ga + r * 136, v[r]);
            }
            v[15] = pk_cmul(v[15], tw2[14]);
            lds_wr(ga + 14 * 136, v[14]);
            lds_wr(ga + 15 * 136, v[15]);
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) v[vv] = lds_rd(xr + vv * 8);
            pk_dft16(v);                                            // v[w] = Z[r + 16 w]
#pragma unroll
            for (int w = 0; w < 16; ++w) lds_wr(ga + w * 128, v[w]);
        } else {
            lds_wr(xa[0], v[0]);
#pragma unroll
            for (int r = 1; r < 15; r += 2) {
                pk_cmul2(v[r], tw2[r - 1], v[r + 1], tw2[r]);
                lds_wr(xa[r & (L - 1)] + r * (8 * L), v[r]);
                lds_wr(xa[(r + 1) & (L - 1)] + (r + 1) * (8 * L), v[r + 1]);
            }
            v[15] = pk_cmul(v[15], tw2[14]);
            lds_wr(xa[L - 1] + 15 * (8 * L), v[15]);
            if constexpr (L == 8) {
                pk2 u0[8], u1[8];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) {
                    u0[pp] = lds_rd(xa[pp] + lp * 64);
                    u1[pp] = lds_rd(xa[pp] + lp * 64 + 512);
                }
                pk_dft8(u0);                                        // u0[w] = Z[lp + 16 w]
                pk_dft8(u1);                                        // u1[w] = Z[lp + 8 + 16 w]
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    lds_wr(ga + w * 128, u0[w]);
                    lds_wr(ga + 64 + w * 128, u1[w]);
                }
            } else {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    pk2 u0 = lds_rd(xa[0] + lp * 32 + hh * 128), u1 = lds_rd(xa[1] + lp * 32 + hh * 128);
                    pk2 u2 = lds_rd(xa[2] + lp * 32 + hh * 128), u3 = lds_rd(xa[3] + lp * 32 + hh * 128);
                    pk_dft4(u0, u1, u2, u3);                        // u_w = Z[lp + 4 hh + 16 w]
                    v[4 * hh] = u0;
                    v[4 * hh + 1] = u1;
                    v[4 * hh + 2] = u2;
                    v[4 * hh + 3] = u3;
                }
#pragma unroll
                for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                    for (int w = 0; w < 4; ++w) lds_wr(ga + hh * 32 + w * 128, v[4 * hh + w]);
            }
        }
        __syncthreads();                                            // C
        // ---- 3. conjugate-symmetric unpack of the pairs (k, M - k) ----------------------------------------------------------
        float* row = (float*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);
        uint32_t* prow = (uint32_t*)row;
        auto zlo = [&](int c, int g2) -> pk2 { return lds_rd((L == 16 ? ulo[0] + c * (RS * 8) : ulo[L == 16 ? 0 : c]) + g2 * (M / 8)); };
        auto zhi = [&](int c, int g2) -> pk2 { return lds_rd(uhi[c] - g2 * (M / 8)); };
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
        pk2 za[2][4], zb[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { za[0][c] = zlo(c, 0); zb[0][c] = zhi(c, 0); }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            if (g2 == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { za[1][c] = zlo(c, 1); zb[1][c] = zhi(c, 1); }
                zb[0][0] = t == 0 ? za[0][0] : zb[0][0];            // Z[M] = Z[0]
            }
            float plo[4], phi[4];
            {
                float pw[4];
                pair_powers2(za[g2][0], zb[g2][0], twur[4 * g2], za[g2][1], zb[g2][1], twur[4 * g2 + 1], pw);
                plo[0] = pw[0]; phi[0] = pw[1]; plo[1] = pw[2]; phi[1] = pw[3];
                pair_powers2(za[g2][2], zb[g2][2], twur[4 * g2 + 2], za[g2][3], zb[g2][3], twur[4 * g2 + 3], pw);
                plo[2] = pw[0]; phi[2] = pw[1]; plo[3] = pw[2]; phi[3] = pw[3];
            }
            const int k0 = 4 * t + (M / 4) * g2;
            if constexpr (KIND == 0) {
                pk_row_store<kNtRows>((pk_f4*)(row + k0), pk_f4{plo[0], plo[1], plo[2], plo[3]});
                pk_row_store<kNtRows>((pk_f4*)(row + M - k0 - 3), pk_f4{phi[3], phi[2], phi[1], phi[0]});
            } else if constexpr (IMAGE) {
                float vv[8];
                uint32_t cc[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    vv[c] = index_value(plo[c], wgr[4 * g2 + c]);
                    vv[4 + c] = index_value(phi[c], wgr[8 + 4 * g2 + c]);
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
                        const int kk = e < 4 ? k0 + e : M - k0 - (e - 4);
                        const int n = exact_colour_index(near_edge, e < 4 ? plo[e] : phi[e - 4], kk, (int)vv[e], a);
                        if (near_edge) cc[e] = lut_lds[n];
                    }
                }
                pk_row_store<kNtRows>((pk_u4*)(prow + k0), pk_u4{cc[0], cc[1], cc[2], cc[3]});
                pk_row_store<kNtRows>((pk_u4*)(prow + M - k0 - 3), pk_u4{cc[7], cc[6], cc[5], cc[4]});
            } else {
                *(pk_f4*)(row + k0) = pk_f4{finish(plo[0], wgr[4 * g2]), finish(plo[1], wgr[4 * g2 + 1]), finish(plo[2], wgr[4 * g2 + 2]),
                                            finish(plo[3], wgr[4 * g2 + 3])};
                *(pk_f4*)(row + M - k0 - 3) = pk_f4{finish(phi[3], wgr[8 + 4 * g2 + 3]), finish(phi[2], wgr[8 + 4 * g2 + 2]),
                                                    finish(phi[1], wgr[8 + 4 * g2 + 1]), finish(phi[0], wgr[8 + 4 * g2])};
            }
        }
        if (t == 0) {
            const pk2 zm = lds_rd(sm + (M / 32) * 8);      // Z[M/2]: region 0 (base 0), slot M/32
            const float pm = (zm.x * zm.x + zm.y * zm.y) * 4.f;
            if constexpr (KIND == 0) {
                row[M / 2] = pm;
            } else if constexpr (IMAGE) {
                const float vv = index_value(pm, wg_nyq);
                int idx = (int)vv;
                const bool near_edge = __builtin_amdgcn_fractf(vv) < a.edge2;
                if (near_edge) idx = exact_colour_index(near_edge, pm, M / 2, idx, a);
                prow[M / 2] = lut_lds[idx];
            } else {
                row[M / 2] = finish(pm, wg_nyq);
            }
        }
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
