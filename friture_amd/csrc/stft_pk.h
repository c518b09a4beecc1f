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

// slot of element e inside a 512-element region
__device__ __forceinline__ int pk_sigma(int e) { return e ^ ((e >> 4) & 7) ^ (((e >> 6) & 1) << 3); }

// KIND: 0 PSD, 1 dB / normalised (run-time choice), 3 colour image.  HS: ring slots (of 512 complex) a hop advances: 8 = hop N/2, 4 = hop N/4.
template <int KIND, int HS>
__global__ void __launch_bounds__(PkPlan::BLOCK, 2) stft_pk_kernel(const StftArgs a) {
    using P = PkPlan;
    constexpr int M = P::M, MS = P::MS, RS = P::RS;
    constexpr int PH = 16 / HS;                                     // frames until the ring is back in phase
    __shared__ __attribute__((aligned(1024))) char smem[P::LDS_BYTES];
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the block
    uint32_t* const lut_lds = (uint32_t*)(smem + P::LUT_OFF);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if constexpr (KIND == 3) {
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
    const float* wgt = (const float*)(KIND == 3 ? a.wimage : a.weight);
    const float image_gain = (float)a.image_gain, norm_off = (float)a.norm_off, norm_scale = (float)a.norm_scale;

    // ---- per-thread constants of a run, in registers ------------------------------------------------------------------
    pk2 winr[16], tw1[15], twur[8], twp1[7], twp2[7];
    float wgr[KIND == 0 ? 1 : 16];
#pragma unroll
    for (int j = 0; j < 16; ++j) winr[j] = win[t + j * MS];
#pragma unroll
    for (int k0 = 1; k0 < 16; ++k0) tw1[k0 - 1] = tw[(t * k0) & (M - 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q) twur[q] = twn[t + q * MS];
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
            wgr[q] = wgt ? wgt[t + q * MS] : 0.f;
            wgr[8 + q] = wgt ? wgt[M - t - q * MS] : 0.f;
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
    // unpack: Z[t + 512 q] and Z[M - t - 512 q] (lds_swizzle_model.py derives these forms)
    // lo: region t & 15, slot sigma((t >> 4) + 32 q) = 32 q + (sigma5(t >> 4) ^ c(q)); hi with tm = 512 - t and q' = 15 - q;
    // c(q) = ((q & 3) << 1) | (((q >> 1) & 1) << 3) takes four values: four byte bases per side, the rest is an immediate.
    // Thread 0's mirror values are Z[512 (16 - q)] (region 0, slot 32 (16 - q)): its four bases are preset so that the
    // same immediates land there; its Z[M] = Z[0] is patched in the frame.
    uint32_t blo[4], bhi[4];
    {
        const int tm = MS - t;                                      // 1..512
        const int ulo = (t & 15) * RS, ulo5 = (t >> 4) ^ ((t >> 8) & 1);
        const int uhi = (tm & 15) * RS, uhi5 = (tm >> 4) ^ ((tm >> 8) & 1);
        const int cls[4] = {0, 2, 12, 14}, lane0[4] = {34, 44, 46, 32};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            blo[c] = sm + (ulo + (ulo5 ^ cls[c])) * 8;
            bhi[c] = sm + (t == 0 ? lane0[c] : uhi + (uhi5 ^ cls[c])) * 8;
        }
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

    // ---- one frame; ph = (frame index inside the run) mod PH is a compile-time constant -------------------------------------
    auto frame = [&](auto phc, int g) -> bool {
        constexpr int ph = decltype(phc)::value;
        if (g >= nfr) return false;
        // the copy of this frame's new samples has landed once at most the 16 row stores of the previous frame (issued
        // after it) are outstanding — vector-memory operations retire in order
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        // ---- 1. samples from the wave's ring slots, window, 16-point DFT over j, twiddle ---------------------------------
        pk2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = lds_rd(ring_lane + ((j + ph * HS) & 15) * 512) * winr[j];
        // every lane of this wave has its samples (the products above exist): the oldest HS slots are free for the next
        // frame's new samples — which are the next frame's slots 16 - HS .. 15
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
        if (g + 1 < nfr) copy_slots(f0 + g + 1, 16 - HS, HS, ph * HS);
        pk_dft16(v);
#pragma unroll
        for (int k0 = 1; k0 < 15; k0 += 2) pk_cmul2(v[k0], tw1[k0 - 1], v[k0 + 1], tw1[k0]);
        v[15] = pk_cmul(v[15], tw1[14]);
        __syncthreads();                                            // A: the previous frame's unpack has read the regions
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) lds_wr(tr_lane + k0 * (RS * 8), v[k0]);
        __syncthreads();                                            // B
        // ---- 2. sixteen 512-point transforms over t: this wave's regions `wave` and 8 + wave, interleaved -------------------
        pk2 u[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) u[r][j] = lds_rd(((j & 1) ? g1 : g0) + r * (8 * RS * 8) + j * 512);
#pragma unroll
        for (int r = 0; r < 2; ++r) pk_dft8(u[r]);
        pass_sync<true>();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t p = sub + ((s0 ^ q) << 3);
            lds_wr(p, u[0][q]);
            lds_wr(p + 8 * RS * 8, u[1][q]);
        }
        pass_sync<true>();
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) u[r][j] = lds_rd(((j & 1) ? g1 : g0) + r * (8 * RS * 8) + j * 512);
#pragma unroll
        for (int q = 1; q < 8; ++q) pk_cmul2(u[0][q], twp1[q - 1], u[1][q], twp1[q - 1]);      // the two rounds share their factors
#pragma unroll
        for (int r = 0; r < 2; ++r) pk_dft8(u[r]);
        pass_sync<true>();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t p = sub + ((s1 ^ (((q >> 1) & 3) | ((q & 1) << 3))) << 3) + (q & ~1) * 64;
            lds_wr(p, u[0][q]);
            lds_wr(p + 8 * RS * 8, u[1][q]);
        }
        pass_sync<true>();
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) u[r][j] = lds_rd(((j & 1) ? g1 : g0) + r * (8 * RS * 8) + j * 512);
#pragma unroll
        for (int q = 1; q < 8; ++q) pk_cmul2(u[0][q], twp2[q - 1], u[1][q], twp2[q - 1]);      // the two rounds share their factors
#pragma unroll
        for (int r = 0; r < 2; ++r) pk_dft8(u[r]);
        pass_sync<true>();
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) lds_wr(((j & 1) ? g1 : g0) + r * (8 * RS * 8) + j * 512, u[r][j]);
        __syncthreads();                                            // C
        // ---- 3. conjugate-symmetric unpack of the pairs (k, M - k), k = t + 512 q ---------------------------------------------
        float* row = (float*)a.out + chan * a.out_cstride + (f0 + g) * (M + 1);
        uint32_t* prow = (uint32_t*)row;
        auto zlo = [&](int q) -> pk2 { return lds_rd(blo[q & 3] + 32 * q * 8); };
        auto zhi = [&](int q) -> pk2 { return lds_rd(bhi[(15 - q) & 3] + 32 * (15 - q) * 8); };
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
        auto image4 = [&](int q, const int (&k)[4], const float (&pw)[4], const float (&w)[4]) {
            float vv[4];
            uint32_t c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vv[e] = clamp_index(image_gain * log2_t(pw[e] + 1e-30f) + w[e]);
                c[e] = lut_lds[(int)vv[e]];
            }
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
            prow[t + q * MS] = c[0];
            prow[M - t - q * MS] = c[1];
            prow[t + (q + 1) * MS] = c[2];
            prow[M - t - (q + 1) * MS] = c[3];
        };
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            pk2 za0 = zlo(q), zb0 = zhi(q), za1 = zlo(q + 1), zb1 = zhi(q + 1);
            if (q == 0) zb0 = t == 0 ? za0 : zb0;                   // Z[M] = Z[0]
            float pw[4];
            pair_powers2(za0, zb0, twur[q], za1, zb1, twur[q + 1], pw);
            if constexpr (KIND == 0) {
                row[t + q * MS] = pw[0];
                row[M - t - q * MS] = pw[1];
                row[t + (q + 1) * MS] = pw[2];
                row[M - t - (q + 1) * MS] = pw[3];
            } else if constexpr (KIND == 3) {
                const int k[4] = {t + q * MS, M - t - q * MS, t + (q + 1) * MS, M - t - (q + 1) * MS};
                const float w[4] = {wgr[q], wgr[8 + q], wgr[q + 1], wgr[8 + q + 1]};
                image4(q, k, pw, w);
            } else {
                row[t + q * MS] = finish(pw[0], wgr[q]);
                row[M - t - q * MS] = finish(pw[1], wgr[8 + q]);
                row[t + (q + 1) * MS] = finish(pw[2], wgr[q + 1]);
                row[M - t - (q + 1) * MS] = finish(pw[3], wgr[8 + q + 1]);
            }
        }
        if (t == 0) {
            const pk2 zm = lds_rd(sm + 256 * 8);           // Z[M/2]: region 0, slot sigma(256) = 256
            const float pm = (zm.x * zm.x + zm.y * zm.y) * 4.f;
            if constexpr (KIND == 0) {
                row[M / 2] = pm;
            } else if constexpr (KIND == 3) {
                const float vv = clamp_index(image_gain * log2_t(pm + 1e-30f) + wg_nyq);
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
