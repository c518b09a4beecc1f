// stft_pk.h — K1 for N = 16384 (BASELINE configs[3]), float32, hop N/2 or N/4, rows on 16-byte boundaries.
// Included by stft.hip after stft_big.h, whose radix-16 x wave-local factorisation it keeps (M = N/2 = 16 * 512:
// thread t of 512 takes the 16-point DFT over z[t + 512 j], twiddles, transposes through LDS; the sixteen 512-point
// transforms over t are wave-local; the unpack reads Z[k] = region k mod 16, slot k / 16).  What is new against
// stft_big_kernel<float, 13, true> (round 3: 0.33 / 0.38 of the HBM peak, 248 registers, 140 KB of LDS,
// SQ_LDS_BANK_CONFLICT = 35 % of the LDS cycles, every sample read twice):
//
//   1. Packed arithmetic.  A wavefront issues one instruction per ~4.5 cycles whatever it is,
//      and with two waves per SIMD this kernel is bound by that, not by the vector pipe.  A complex value lives in an
//      aligned register pair and every butterfly add, twiddle product and window multiply is ONE v_pk_*_f32 — the
//      multiplications by -i / +i are the instruction's op_sel / neg modifiers (inline assembly: the compiler spends a
//      v_xor + v_mov on each).  ~430 vector instructions of arithmetic per frame and thread instead of ~830.
//   2. XOR-swizzled LDS regions instead of padded ones.  Padding made the stride-8 scatters of the sub-transform
//      passes conflict-free and every contiguous access twice as slow.  Slot e of a region lives at
//      sigma(e) = e ^ ((e >> 4) & 7) ^ (((e >> 6) & 1) << 3): contiguous runs stay permutations of their 16 / 32 banks,
//      the scatters of both passes spread over all 16, and with a region stride of 514 the unpack's sixteen-region
//      reads do too (tools/exp/lds_swizzle_model.py: 3200 LDS-array cycles per frame against 5248; 3072 is the floor).
//   3. The frame's samples stay in an LDS ring; only the hop's NEW samples are copied per frame (global_load_lds,
//      one half or one quarter of a frame), into the slots of the oldest ones.  Every wavefront copies exactly the
//      samples its own threads read (lanes 0-31 / 32-63 of a copy gather two 512-byte pieces), so no barrier stands
//      between a copy and its use — only the wave's own hand-counted vmcnt — and the first stage of the NEXT frame
//      runs before the barrier that ends this frame's unpack.
//   4. Both rounds of eight sub-transforms run interleaved in one instruction stream (the first stage's registers are
//      free by then): three exposed LDS round trips per frame instead of six.
//   5. The output kind is a template parameter (the PSD kind keeps no weights in registers).
#pragma once

#ifndef FRT_PK_TIMING          // experiment builds only (wrong output by design): per-wave cycle counts of the frame's eight
#define FRT_PK_TIMING 0        // intervals, written over the first row of every run (tools/exp/pk_timing.py reads them)
#endif
#ifndef FRT_PK_TW_AFTER        // 1: the first stage's twiddle products are issued after barrier A, between the transpose writes
#define FRT_PK_TW_AFTER 1      // (the phase is bound by the LDS write path, its vector pipe is idle: +1-2 % at hop N/4, equal at N/2)
#endif
#ifndef FRT_PK_XREG            // 1: the exchange between the second and third pass of a sub-transform is a register transpose
#define FRT_PK_XREG 0          // (v_permlane32_swap, v_permlane16_swap, masked DPP row_ror:8 moves) instead of an LDS round trip.
#endif                         // Parity-green and measured equal (PSD -1 %, colour +0.7 %): it moves 1100 LDS-pipe cycles per frame
                               // to ~950 cycles on each SIMD's vector pipe (a permlane swap issues every 8.3 cycles) — off
#ifndef FRT_PK_OVERLAP         // 1: (PSD kind) the first stage of frame g + 1 is issued inside frame g's sub-transform phase.  Parity-
#define FRT_PK_OVERLAP 0       // green, measured: the phase grows by 1150-1380 cycles for the 670 moved into it (8500 against 8200 cycles
#endif                         // per frame): that phase has no idle issue slots to give — off
#ifndef FRT_PK_STAGGER         // 1: the two rounds of sub-transforms half a pass apart (software pipeline); 0: in lock step
#define FRT_PK_STAGGER 1
#endif
#ifndef FRT_PK_PRIO            // 1: waves 4-7 (the later-dispatched half) run at s_setprio 1; 2: the two halves of the workgroup
#define FRT_PK_PRIO 0          // (the older and the younger wave of every SIMD) swap priority at every step of the frame
#endif

namespace frt {

typedef float pk2 __attribute__((ext_vector_type(2)));

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ pk2 pk_add_mi(pk2 a, pk2 b) {
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - (-i) b = a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ pk2 pk_sub_mi(pk2 a, pk2 b) {
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ pk2 pk_add_conj(pk2 a, pk2 b) {
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ pk2 pk_sub_conj(pk2 a, pk2 b) {
    pk2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (-i) a = (a.y, -a.x)
__device__ __forceinline__ pk2 pk_mul_mi(pk2 a) {
    pk2 r;
    asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
// a * w: (a.x w.x - a.y w.y, a.x w.y + a.y w.x) — two instructions in ONE asm statement: between two dependent asm
// statements the compiler's hazard recogniser puts an s_nop (an issue slot; 70 per frame in the first build of this file).
// w in vector registers / in a scalar register pair (compile-time constants).
__device__ __forceinline__ pk2 pk_cmul(pk2 a, pk2 w) {
    pk2 t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ pk2 pk_cmul_s(pk2 a, pk2 w) {
    pk2 t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=v"(r), "=&v"(t) : "v"(a), "s"(w));
    return r;
}
// two independent products, their instructions interleaved (the fma of a product waits for its mul: the other product's
// instructions fill the gap)
__device__ __forceinline__ void pk_cmul2(pk2& a0, pk2 w0, pk2& a1, pk2 w1) {
    pk2 t0, t1, r0, r1;
    asm("v_pk_mul_f32 %2, %4, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_mul_f32 %3, %6, %7 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_fma_f32 %0, %4, %5, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %6, %7, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(w0), "v"(a1), "v"(w1));
    a0 = r0;
    a1 = r1;
}
__device__ __forceinline__ void pk_cmul2_s(pk2& a0, pk2 w0, pk2& a1, pk2 w1) {
    pk2 t0, t1, r0, r1;
    asm("v_pk_mul_f32 %2, %4, %5 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_mul_f32 %3, %6, %7 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\t"
        "v_pk_fma_f32 %0, %4, %5, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %6, %7, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1) : "v"(a0), "s"(w0), "v"(a1), "s"(w1));
    a0 = r0;
    a1 = r1;
}

// ---- 8 x 8 transpose between a lane's eight registers and lane bits 3-5 ---------------------------------------------------
// After the second radix-8 pass lane i = 8 ih + il of a sub-transform holds the elements 64 ih + 8 q + il (q = register);
// the third pass wants lane 8 q + il to hold 64 j + 8 q + il in register j: registers and the upper three lane bits trade
// places, il stays.  Three butterfly stages, one per bit; the sub-transform phase is bound by LDS instruction throughput
// (4260 cycles for 96 LDS instructions per wave, whatever their order), the vector pipe has room.
typedef unsigned pk_u2 __attribute__((ext_vector_type(2)));
// a[lanes 32-63] <-> b[lanes 0-31]
__device__ __forceinline__ void pk_swap32(pk2& a, pk2& b) {
    const pk_u2 r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const pk_u2 r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = pk2{__uint_as_float(r0.x), __uint_as_float(r1.x)};
    b = pk2{__uint_as_float(r0.y), __uint_as_float(r1.y)};
}
// a[rows 1, 3] <-> b[rows 0, 2] (rows of 16 lanes)
__device__ __forceinline__ void pk_swap16(pk2& a, pk2& b) {
    const pk_u2 r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const pk_u2 r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = pk2{__uint_as_float(r0.x), __uint_as_float(r1.x)};
    b = pk2{__uint_as_float(r0.y), __uint_as_float(r1.y)};
}
// a[lanes with bit 3 set] <-> b[lanes with bit 3 clear], partner lane ^ 8: DPP moves (row_ror:8) whose bank mask writes only
// the upper / lower half of every row of 16; the unwritten lanes keep `old`.  (v_cndmask_b32 on a vcc that a scalar
// instruction wrote runs at 23 cycles per instruction on this chip — tools/exp/pk_pipes.cpp — a masked DPP move at 4.4.)
__device__ __forceinline__ float pk_dpp_ror8(float old, float src, int bank_mask_hi) {
    return bank_mask_hi ? __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(src), 0x128, 0xf, 0xc, false))
                        : __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(src), 0x128, 0xf, 0x3, false));
}
__device__ __forceinline__ void pk_swap8(pk2& a, pk2& b) {
    const pk2 na = {pk_dpp_ror8(a.x, b.x, 1), pk_dpp_ror8(a.y, b.y, 1)};      // lanes 8-15 of every row take b[lane ^ 8]
    const pk2 nb = {pk_dpp_ror8(b.x, a.x, 0), pk_dpp_ror8(b.y, a.y, 0)};      // lanes 0-7 take a[lane ^ 8]
    a = na;
    b = nb;
}
__device__ __forceinline__ void pk_lane_transpose(pk2 (&u)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pk_swap32(u[r], u[r + 4]);         // lane bit 5 <-> register bit 2
    pk_swap16(u[0], u[2]);                                           // lane bit 4 <-> register bit 1
    pk_swap16(u[1], u[3]);
    pk_swap16(u[4], u[6]);
    pk_swap16(u[5], u[7]);
#pragma unroll
    for (int r = 0; r < 8; r += 2) pk_swap8(u[r], u[r + 1]);        // lane bit 3 <-> register bit 0
}

__device__ __forceinline__ void pk_dft4(pk2& a0, pk2& a1, pk2& a2, pk2& a3) {
    const pk2 s0 = a0 + a2, s1 = a0 - a2, s2 = a1 + a3, d = a1 - a3;
    a0 = s0 + s2;
    a2 = s0 - s2;
    a1 = pk_add_mi(s1, d);
    a3 = pk_sub_mi(s1, d);
}

// forward 8-point DFT, natural order in and out: 28 packed instructions
__device__ __forceinline__ void pk_dft8(pk2 (&a)[8]) {
    const pk2 hm = {0.70710678118654752440f, -0.70710678118654752440f};      // W8^1 = h (1 - i)
    const pk2 hmm = {-0.70710678118654752440f, -0.70710678118654752440f};    // W8^3 = -h (1 + i)
    pk2 e0 = a[0] + a[4], e1 = a[1] + a[5], e2 = a[2] + a[6], e3 = a[3] + a[7];
    pk2 d0 = a[0] - a[4], d1 = a[1] - a[5], d2 = a[2] - a[6], d3 = a[3] - a[7];
    pk2 o1 = d1, o3 = d3;
    pk_cmul2_s(o1, hm, o3, hmm);
    pk_dft4(e0, e1, e2, e3);
    // DFT4 of (d0, o1, -i d2, o3) with the -i folded into the first butterflies
    const pk2 s0 = pk_add_mi(d0, d2), s1 = pk_sub_mi(d0, d2), s2 = o1 + o3, d = o1 - o3;
    a[0] = e0; a[2] = e1; a[4] = e2; a[6] = e3;
    a[1] = s0 + s2;
    a[5] = s0 - s2;
    a[3] = pk_add_mi(s1, d);
    a[7] = pk_sub_mi(s1, d);
}

// forward 16-point DFT of a[j], natural order in and out (n = m + 4 p, k = q + 4 r): 83 packed instructions
__device__ __forceinline__ void pk_dft16(pk2 (&a)[16]) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    const pk2 w1 = {c1, -s1}, w2 = {h, -h}, w3 = {s1, -c1}, w6 = {-h, -h}, w9 = {-c1, s1};
#pragma unroll
    for (int m = 0; m < 4; ++m) pk_dft4(a[m], a[m + 4], a[m + 8], a[m + 12]);      // a[m + 4 q] = sum_p a[m + 4 p] W4^{pq}
    pk_cmul2_s(a[5], w1, a[9], w2);
    pk_cmul2_s(a[13], w3, a[6], w2);
    pk_cmul2_s(a[14], w6, a[7], w3);
    pk_cmul2_s(a[11], w6, a[15], w9);
    a[10] = pk_mul_mi(a[10]);
#pragma unroll
    for (int q = 0; q < 4; ++q) pk_dft4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);      // a[4 q + r] = X[q + 4 r]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = q + 1; r < 4; ++r) {
            const pk2 tmp = a[4 * q + r];
            a[4 * q + r] = a[4 * r + q];
            a[4 * r + q] = tmp;
        }
}

struct PkPlan {
    static constexpr int LOG2M = 13, M = 1 << LOG2M, MS = M / 16, BLOCK = MS, NW = MS / 64;
    static constexpr int RS = MS + 2;                               // region stride (complex): the unpack's 16-region reads hit 32 banks
    static constexpr int REG_BYTES = 16 * RS * 8;                   // 65 792
    static constexpr int RING_OFF = (REG_BYTES + 1023) / 1024 * 1024;
    static constexpr int LUT_OFF = RING_OFF + M * 8;
    static constexpr int LDS_BYTES = LUT_OFF + 1024;                // 133 120
};

// LDS accesses by 32-bit byte address, volatile: the compiler's load/store optimiser otherwise pairs them into ds_read2st64_b64 /
// ds_write2st64_b64, which move the same bytes in twice the LDS cycles (MI355X_MICROARCH.md §LDS: 128 against 256 B/clk for reads)
typedef __attribute__((address_space(3))) pk2 lds_pk2;
__device__ __forceinline__ pk2 lds_rd(uint32_t addr) { return *(const volatile lds_pk2*)addr; }
__device__ __forceinline__ void lds_wr(uint32_t addr, pk2 v) { *(volatile lds_pk2*)addr = v; }

// a 16-byte row store on a 4-byte boundary (rows of N/2 + 1 values), non-temporal or plain
template <bool NT, typename V>
__device__ __forceinline__ void pk_row_store(void* p, V v) {
    typedef V row_vec __attribute__((aligned(4)));
    if constexpr (NT) __builtin_nontemporal_store(v, (row_vec*)p);
    else *(row_vec*)p = v;
}

#ifdef FRT_EXPERIMENTS   // round 3's kernel: superseded by stft_pk16_kernel (stft_pk16.h), kept for A/B builds of tools/exp
// slot of element e inside a 512-element region
__device__ __forceinline__ int pk_sigma(int e) { return e ^ ((e >> 4) & 7) ^ (((e >> 6) & 1) << 3); }

// KIND: 0 PSD, 1 dB / normalised (run-time choice), 3 colour image, 4 colour image without the + 1e-30 (StftArgs::eps_free).  HS: ring slots (of 512 complex) a hop advances: 8 = hop N/2, 4 = hop N/4.
template <int KIND, int HS>
__global__ void __launch_bounds__(PkPlan::BLOCK, 2) stft_pk_kernel(const StftArgs a) {
    using P = PkPlan;
    constexpr int M = P::M, MS = P::MS, RS = P::RS;
    constexpr int PH = 16 / HS;                                     // frames until the ring is back in phase
    constexpr bool IMAGE = KIND >= 3, EPS_FREE = KIND == 4;
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

    // ---- per-thread constants of a run, in registers ------------------------------------------------------------------
    pk2 winr[16], tw1[15], twur[8], twp1[7], twp2[7];
    float wgr[KIND == 0 ? 1 : 16];
#pragma unroll
    for (int j = 0; j < 16; ++j) winr[j] = win[t + j * MS];
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) tw1[k0 - 1] = tw[(t * k0) & (M - 1)];
    // unpack: thread t owns the bin pairs (k, M - k), k = 4 t + c + 2048 g (c < 4, g < 2): four consecutive bins per 16-byte store
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) twur[4 * g + c] = twn[4 * t + c + 2048 * g];
    // sub-transforms of 512 points, 8 per lane: pass 1 multiplies slot q by W_64^{q (lane & 7)}, pass 2 by W_512^{q lane}
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        twp1[q - 1] = tws[(q * (lane & 7) * 8) & (MS - 1)];
        twp2[q - 1] = tws[(q * lane) & (MS - 1)];
    }
    float wg_nyq = 0.f;
    if constexpr (KIND != 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 4 * t + (q & 3) + 2048 * (q >> 2);
            wgr[q] = wgt ? wgt[k] : 0.f;
            wgr[8 + q] = wgt ? wgt[M - k] : 0.f;
        }
        wg_nyq = wgt ? wgt[M / 2] : 0.f;
    }

    // ---- LDS addresses (bytes) -------------------------------------------------------------------------------------------
    // ring: [wave][slot 0..15][lane], one slot = the wave's 64 complex samples z[64 wave + lane + 512 j]
    const uint32_t ring_wave = sm + (uint32_t)P::RING_OFF + (uint32_t)wave * 8192u;
    const uint32_t ring_lane = ring_wave + lane * 8;
    // transpose: region k0, slot sigma(t)
    const uint32_t tr_lane = sm + pk_sigma(t) * 8;
    // sub-transforms: region `wave` (round 0) and 8 + wave (round 1: + 8 RS), lane = index inside the sub-transform
    const int a0 = lane ^ ((lane >> 4) & 3);                       // natural position lane + 64 j -> 64 j + (a0 ^ 12 (j & 1))
    const uint32_t sub = sm + wave * (RS * 8);
    const uint32_t g0 = sub + a0 * 8;
    const uint32_t g1 = sub + (a0 ^ 12) * 8;
    const int s0 = pk_sigma(8 * lane);                              // pass-0 scatter: element 8 lane + q -> s0 ^ q
    const int b3 = (lane >> 3) & 1;
    const int s1 = (((lane >> 3) << 6) | (lane & 7)) ^ (12 * b3);   // pass-1 scatter: element 64 (lane >> 3) + (lane & 7) + 8 q
    // unpack: Z[k] = region k & 15, slot sigma(k >> 4).  For k = 4 t + c + 2048 g: region 4 (t & 3) + c, slot sigma(t >> 2) + 128 g
    // (bit 7 of a slot is outside the swizzle): one lane base, c and g are immediates.  Z[M - k]: with u = 4 t + c, region
    // (16 - (u & 15)) & 15 and slot 512 - 128 g - ((u + 15) >> 4): one lane base per c.  (u = 0: g = 0 reads the pad slot 512
    // and is replaced by Z[0], g = 1 lands on Z[6144] = region 0, slot 384.)
    const uint32_t ulo = sm + ((4 * (t & 3)) * RS + pk_sigma(t >> 2)) * 8;
    uint32_t uhi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int u = 4 * t + c, sl = 512 - ((u + 15) >> 4);
        uhi[c] = sm + (((16 - (u & 15)) & 15) * RS + (sl < 512 ? pk_sigma(sl) : 512)) * 8;
    }

    // ---- sample copies ----------------------------------------------------------------------------------------------------
    // one copy instruction = 1 KB = ring slots (s, s + 1) of this wave: lanes 0-31 fetch z[64 wave .. + 64) + 512 j, lanes 32-63 the same at j + 1
    const uint32_t copy_lane = (uint32_t)(lane & 31) * 16u + (uint32_t)(lane >> 5) * 4096u;
    auto copy_slots = [&](long long frame, int j_first, int n_slots, int ring_slot_first) {
        // frame's slots j_first .. j_first + n_slots - 1 -> ring slots ring_slot_first ..
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
#if FRT_PK_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#if FRT_PK_PRIO == 2
    // The issue arbiter serves the older wave of a SIMD first (waves 0-3 here): measured with FRT_PK_TIMING, the older half
    // finishes the sub-transform phase in 2940 cycles, the younger in 4315, and the older then idles at the barrier while the
    // younger runs alone.  Swapping the priority at every step lets both halves progress at the same average rate.
    const bool wave_hi = wave >= 4;
#define PK_STEP(s)                                                         \
    do {                                                                   \
        if (wave_hi == (((s) & 1) != 0)) __builtin_amdgcn_s_setprio(1);    \
        else __builtin_amdgcn_s_setprio(0);                                \
    } while (0)
#else
#define PK_STEP(s) do { } while (0)
#endif

    // ---- first stage of frame g (ph = g mod PH, compile time): samples from the wave's ring slots, window, 16-point DFT over j
    // (and the twiddles unless they ride between the transpose writes) ---------------------------------------------------------
    // OVL (experiment, PSD kind: it has the 32 registers to keep the result alive through the unpack): the first stage of frame
    // g + 1 issued INSIDE frame g's sub-transform phase instead of between barrier C of one frame and barrier A of the next.
    constexpr bool OVL = FRT_PK_OVERLAP && KIND == 0;
    pk2 v[16];                                                    // first-stage output of the frame about to be transposed
    auto first_stage = [&](auto phc, int g) {
        constexpr int ph = decltype(phc)::value;
        // the copy of this frame's new samples has landed once at most the 4 row stores issued after it (wave 0 has a
        // fifth, the bin N/4: it waits for one store more than it must) are outstanding — vector-memory operations retire
        // in order
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = lds_rd(ring_lane + ((j + ph * HS) & 15) * 512) * winr[j];
        // every lane of this wave has its samples (the products above exist): the oldest HS slots are free for the next
        // frame's new samples — which are the next frame's slots 16 - HS .. 15
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
        if (g + 1 < nfr) copy_slots(f0 + g + 1, 16 - HS, HS, ph * HS);
        pk_dft16(v);
#if !FRT_PK_TW_AFTER
#pragma unroll
        for (int k0 = 1; k0 < 15; k0 += 2) pk_cmul2(v[k0], tw1[k0 - 1], v[k0 + 1], tw1[k0]);
        v[15] = pk_cmul(v[15], tw1[14]);
#endif
    };

    // ---- one frame; ph = (frame index inside the run) mod PH is a compile-time constant -------------------------------------
    auto frame = [&](auto phc, int g) -> bool {
        constexpr int ph = decltype(phc)::value;
        if (g >= nfr) return false;
        PK_TICK(-1);
        PK_STEP(0);
        if constexpr (!OVL) first_stage(phc, g);                  // (OVL: issued inside the previous frame, or in front of the loop)
        PK_TICK(0);
        PK_TICK(1);
        __syncthreads();                                            // A: the previous frame's unpack has read the regions
        PK_TICK(2);
        PK_STEP(1);
#if FRT_PK_TW_AFTER
        lds_wr(tr_lane, v[0]);
#pragma unroll
        for (int k0 = 1; k0 < 15; k0 += 2) {
            pk_cmul2(v[k0], tw1[k0 - 1], v[k0 + 1], tw1[k0]);
            lds_wr(tr_lane + k0 * (RS * 8), v[k0]);
            lds_wr(tr_lane + (k0 + 1) * (RS * 8), v[k0 + 1]);
        }
        v[15] = pk_cmul(v[15], tw1[14]);
        lds_wr(tr_lane + 15 * (RS * 8), v[15]);
#else
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) lds_wr(tr_lane + k0 * (RS * 8), v[k0]);
#endif
        PK_TICK(3);
        __syncthreads();                                            // B
        PK_TICK(4);
        PK_STEP(2);
        // ---- 2. sixteen 512-point transforms over t: this wave's regions `wave` (round 0) and 8 + wave (round 1) ------------------
        // LDS traffic of a wave is executed in order and the accesses are volatile (program order), so the scatter of a pass
        // and the gather behind it need no fence; what the source order decides is which round's arithmetic covers which
        // round's LDS round trip.
        pk2 u[2][8];
        auto gather = [&](int r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) u[r][j] = lds_rd(((j & 1) ? g1 : g0) + r * (8 * RS * 8) + j * 512);
        };
        auto scatter0 = [&](int r) {
#pragma unroll
            for (int q = 0; q < 8; ++q) lds_wr(sub + ((s0 ^ q) << 3) + r * (8 * RS * 8), u[r][q]);
        };
        auto scatter1 = [&](int r) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                lds_wr(sub + ((s1 ^ (((q >> 1) & 3) | ((q & 1) << 3))) << 3) + (q & ~1) * 64 + r * (8 * RS * 8), u[r][q]);
        };
        auto writeback = [&](int r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) lds_wr(((j & 1) ? g1 : g0) + r * (8 * RS * 8) + j * 512, u[r][j]);
        };
#if FRT_PK_STAGGER
        // round 1 runs half a pass behind round 0: while one round's scatter and gather are in flight the other computes
        auto twiddle = [&](int r, const pk2 (&w)[7]) {
#pragma unroll
            for (int q = 1; q < 7; q += 2) pk_cmul2(u[r][q], w[q - 1], u[r][q + 1], w[q]);
            u[r][7] = pk_cmul(u[r][7], w[6]);
        };
        gather(0);
        gather(1);
        pk_dft8(u[0]);
        scatter0(0);
        gather(0);
        pk_dft8(u[1]);
        scatter0(1);
        gather(1);
        if constexpr (OVL) {
            if (g + 1 < nfr) first_stage(std::integral_constant<int, (ph + 1) % PH>{}, g + 1);
        }
        twiddle(0, twp1);
        pk_dft8(u[0]);
#if FRT_PK_XREG
        pk_lane_transpose(u[0]);
        twiddle(1, twp1);
        pk_dft8(u[1]);
        pk_lane_transpose(u[1]);
#else
        scatter1(0);
        gather(0);
        twiddle(1, twp1);
        pk_dft8(u[1]);
        scatter1(1);
        gather(1);
#endif
        twiddle(0, twp2);
        pk_dft8(u[0]);
        writeback(0);
        twiddle(1, twp2);
        pk_dft8(u[1]);
        writeback(1);
#else
        gather(0);
        gather(1);
        pk_dft8(u[0]);
        pk_dft8(u[1]);
        PK_STEP(3);
        scatter0(0);
        scatter0(1);
        PK_STEP(4);
        gather(0);
        gather(1);
        if constexpr (OVL) {
            if (g + 1 < nfr) first_stage(std::integral_constant<int, (ph + 1) % PH>{}, g + 1);
        }
#pragma unroll
        for (int q = 1; q < 8; ++q) pk_cmul2(u[0][q], twp1[q - 1], u[1][q], twp1[q - 1]);      // the two rounds share their factors
        pk_dft8(u[0]);
        pk_dft8(u[1]);
        PK_STEP(5);
#if FRT_PK_XREG
        pk_lane_transpose(u[0]);
        pk_lane_transpose(u[1]);
#else
        scatter1(0);
        scatter1(1);
        PK_STEP(6);
        gather(0);
        gather(1);
#endif
#pragma unroll
        for (int q = 1; q < 8; ++q) pk_cmul2(u[0][q], twp2[q - 1], u[1][q], twp2[q - 1]);
        pk_dft8(u[0]);
        pk_dft8(u[1]);
        PK_STEP(7);
        writeback(0);
        writeback(1);
#endif
        PK_TICK(5);
        __syncthreads();                                            // C
        PK_TICK(6);
        PK_STEP(8);
        // ---- 3. conjugate-symmetric unpack of the pairs (k, M - k), k = t + 512 q ---------------------------------------------
        float* row = (float*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);
        uint32_t* prow = (uint32_t*)row;
        auto zlo = [&](int c, int g) -> pk2 { return lds_rd(ulo + c * (RS * 8) + g * 1024); };
        auto zhi = [&](int c, int g) -> pk2 { return lds_rd(uhi[c] - g * 1024); };
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
        auto finish = [&](float p, float w) -> float {             // dB kinds
            float vv = db10<float>(p) + w;
            if (a.kind == FRT_STFT_NORM) vv = (vv + norm_off) * norm_scale;
            return vv;
        };
        // colour index of a power (exact, see stft_wave.h: exact_colour_index): float32 value, clamped into the LUT's range
        auto index_value = [&](float p, float w) -> float {
            return clamp_index(image_gain * log2_t(EPS_FREE ? p : p + 1e-30f) + w);
        };
        // Two groups (g = 0, 1) of four pairs: bins k0 + c and M - k0 - c, k0 = 4 t + 2048 g — one 16-byte store per side and
        // group (1 KB per wave-instruction; rows start on 4-byte boundaries, the hardware splits what straddles a line): 4
        // vector-memory instructions per thread and frame instead of 16.  The memory-path counters of the N = 1024 kernel
        // (profiles/r04_headline_memory_path_pmc.txt) show the texture addresser busy ~22 cycles per wave-instruction
        // whatever it carries.  The Z values of group 1 are requested before group 0's arithmetic.
        typedef float pk_f4 __attribute__((ext_vector_type(4), aligned(4)));
        typedef uint32_t pk_u4 __attribute__((ext_vector_type(4), aligned(4)));
        pk2 za[2][4], zb[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { za[0][c] = zlo(c, 0); zb[0][c] = zhi(c, 0); }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { za[1][c] = zlo(c, 1); zb[1][c] = zhi(c, 1); }
                zb[0][0] = t == 0 ? za[0][0] : zb[0][0];            // Z[M] = Z[0]
            }
            float plo[4], phi[4];
            {
                float pw[4];
                pair_powers2(za[g][0], zb[g][0], twur[4 * g], za[g][1], zb[g][1], twur[4 * g + 1], pw);
                plo[0] = pw[0]; phi[0] = pw[1]; plo[1] = pw[2]; phi[1] = pw[3];
                pair_powers2(za[g][2], zb[g][2], twur[4 * g + 2], za[g][3], zb[g][3], twur[4 * g + 3], pw);
                plo[2] = pw[0]; phi[2] = pw[1]; plo[3] = pw[2]; phi[3] = pw[3];
            }
            const int k0 = 4 * t + 2048 * g;
            if constexpr (KIND == 0) {
                *(pk_f4*)(row + k0) = pk_f4{plo[0], plo[1], plo[2], plo[3]};
                *(pk_f4*)(row + M - k0 - 3) = pk_f4{phi[3], phi[2], phi[1], phi[0]};
            } else if constexpr (IMAGE) {
                float vv[8];
                uint32_t cc[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    vv[c] = index_value(plo[c], wgr[4 * g + c]);
                    vv[4 + c] = index_value(phi[c], wgr[8 + 4 * g + c]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) cc[e] = lut_lds[(int)vv[e]];
                float m = __builtin_amdgcn_fractf(vv[0]);
#pragma unroll
                for (int e = 1; e < 8; ++e) m = fminf(m, __builtin_amdgcn_fractf(vv[e]));
                if (__any(m < a.edge2)) {                           // within 2 thr above an index edge: one float64 comparison decides
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool near_edge = __builtin_amdgcn_fractf(vv[e]) < a.edge2;
                        const int kk = e < 4 ? k0 + e : M - k0 - (e - 4);
                        const int n = exact_colour_index(near_edge, e < 4 ? plo[e] : phi[e - 4], kk, (int)vv[e], a);
                        if (near_edge) cc[e] = lut_lds[n];
                    }
                }
                *(pk_u4*)(prow + k0) = pk_u4{cc[0], cc[1], cc[2], cc[3]};
                *(pk_u4*)(prow + M - k0 - 3) = pk_u4{cc[7], cc[6], cc[5], cc[4]};
            } else {
                *(pk_f4*)(row + k0) = pk_f4{finish(plo[0], wgr[4 * g]), finish(plo[1], wgr[4 * g + 1]), finish(plo[2], wgr[4 * g + 2]),
                                            finish(plo[3], wgr[4 * g + 3])};
                *(pk_f4*)(row + M - k0 - 3) = pk_f4{finish(phi[3], wgr[8 + 4 * g + 3]), finish(phi[2], wgr[8 + 4 * g + 2]),
                                                    finish(phi[1], wgr[8 + 4 * g + 1]), finish(phi[0], wgr[8 + 4 * g])};
            }
        }
        if (t == 0) {
            const pk2 zm = lds_rd(sm + 256 * 8);           // Z[M/2]: region 0, slot sigma(256) = 256
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
        PK_TICK(7);
        return true;
    };
    if constexpr (OVL) {
        if (nfr > 0) first_stage(std::integral_constant<int, 0>{}, 0);
    }
    for (int g = 0; g < nfr; g += PH) {
        if (!frame(std::integral_constant<int, 0>{}, g)) break;
        if (!frame(std::integral_constant<int, 1>{}, g + 1)) break;
        if constexpr (PH > 2) {
            if (!frame(std::integral_constant<int, 2>{}, g + 2)) break;
            if (!frame(std::integral_constant<int, 3>{}, g + 3)) break;
        }
    }
#if FRT_PK_TIMING
    // interval i of this wave, summed over the run's frames: 0 wait for the copy + ring reads + window, 1 DFT16 (+ twiddles),
    // 2 barrier A, 3 transpose writes, 4 barrier B, 5 sub-transforms, 6 barrier C, 7 unpack + stores
    if (lane == 0 && nfr > 0) {
        float* row = (float*)a.out + chan * a.out_cstride + f0 * (M + 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) row[wave * 8 + i] = (float)tacc[i] / (float)nfr;
        // shader clock over the run: s_memtime ticks per 100 MHz s_memrealtime tick
        row[64 + wave] = (float)(__builtin_amdgcn_s_memtime() - run_t0) / (float)(__builtin_amdgcn_s_memrealtime() - run_r0) * 0.1f;
    }
#endif
#undef PK_TICK
#undef PK_STEP
}

#endif  // FRT_EXPERIMENTS
}  // namespace frt
