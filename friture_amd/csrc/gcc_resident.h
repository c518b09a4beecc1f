// gcc_resident.h — GCC-PHAT of the delay estimator's default window (L = 24000 samples, friture/delay_estimator.py:114-115;
// semantics: friture/signal/correlation.py:24-43) with NOTHING in HBM between the two signals and the correlation.
//
// gcc_phat_kernel<2, 1> (gcc.hip) parks three of a pair's four 96 KB sub-spectra in a per-pair scratch slab and reads them back for
// the cross spectrum: +288 KB written and +288 KB read beside the 576 KB a pair moves algorithmically, and its sample loads take
// every second 16-byte word of a line (sub-transform r takes the sample pairs 2 m + r), four load phases per pair that each fetch
// whole lines for half their bytes.  Here the parked sub-spectra live where the CU has room for them:
//
//   * a CU's register file is 512 KB and this kernel owns it (one workgroup per CU).  The 6000-point transform needs a part of a
//     thread's registers; the others park data across it.
//   * ownership by QUADS of bins: with M = 2 M2 the bins {q, M - q, M2 - q, M2 + q}, q <= M2 / 2, are formed from the SAME two
//     elements q and M2 - q of every sub-spectrum — Z_s[q] = a + t b, Z_s[M2 + q] = a - t b, Z_s[M - q] = c + conj(t) d,
//     Z_s[M2 - q] = c - conj(t) d with (a, b) = S_s0/1[q], (c, d) = S_s0/1[M2 - q], t = exp(-2 pi i q / M) — and so are the four
//     inverse-transform inputs of the quad (elements q and M2 - q of both inverse sub-transforms).  Thread t owns the quads
//     q = t + NT i: it reads its elements of a sub-spectrum out of LDS when the transform ends and nobody else ever needs them —
//     no exchange between threads from the forward transforms to the inverse ones but the block maximum of |G|.
//   * signal 0's two sub-spectra wait in registers; signal 1's first waits in the 64 KB of LDS the transform array leaves free
//     (+ a few registers); its second is read where the transform left it.
//   * a signal is loaded ONCE, whole lines (both 16-byte words of a 32-byte slot in one phase): the r = 0 half goes into the
//     transform array, the r = 1 half waits in the second LDS array (+ a few registers) for the first transform to end.
//   * the r = 0 half of the correlation waits in registers for the r = 1 half, so a thread stores 32 contiguous bytes per slot.
//   * a forward transform's FIRST pass is taken where its input is formed (a thread loads the six points of its own butterflies: no
//     write of the input into the transform array and no gather of it back), an inverse transform's LAST pass ends in the registers
//     the correlation is stored from (no scatter, no read-back): 100 / 1024 pairs 75.6 / 360 -> 71.4 / 335 us.
//
// HBM traffic per pair = the algorithmic 24 L bytes (+ table reads that hit in L2).
#pragma once

namespace frt {

constexpr int kResM2 = 6000, kResM = 2 * kResM2, kResL = 2 * kResM;
constexpr int kResSpare = 4096;                                                   // complex doubles of the second LDS array

template <int NT>
struct ResPlan {
    static constexpr int NS = (kResM2 + NT - 1) / NT;                             // points of a sub-transform per thread (slots)
    static constexpr int NQ = (kResM2 / 2 + 1 + NT - 1) / NT;                     // quads per thread (q <= 3000)
    static constexpr int NE = 2 * NQ;                                             // elements of a sub-spectrum per thread
    static constexpr int SPS = kResSpare / NT;                                    // a thread's slots in the second LDS array
    static constexpr int NW = NT / 64;
    static constexpr size_t LDS_BYTES = (size_t)(kResM2 + SPS * NT) * 16 + 2 * 16 * sizeof(double) + 16 * sizeof(int);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
    static_assert(NW <= 16 && NT % 64 == 0, "reduction arrays");
    static constexpr bool slot_full(int j) { return (j + 1) * NT <= kResM2; }     // every thread's point of slot j exists
    static constexpr bool quad_full(int i) { return (i + 1) * NT <= kResM2 / 2 + 1; }
};

// Buffer accesses: the descriptor in scalar registers, ONE 32-bit lane offset and a scalar offset per access.  As flat accesses the
// loads of a signal and the stores of a correlation each keep a 64-bit address pair alive — computed at the top of the kernel,
// spilled, reloaded.  (A 16-byte buffer access needs 4-byte alignment only: windows that are views into a ring go the same way.)
typedef __amdgpu_buffer_rsrc_t res_rsrc;
typedef double res_d2 __attribute__((ext_vector_type(2)));
typedef uint32_t res_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ res_rsrc res_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);      // raw buffer, gfx950 data format word
}
__device__ __forceinline__ cpx<double> res_load16(res_rsrc r, uint32_t lane_off, uint32_t soff) {
    const res_d2 v = __builtin_bit_cast(res_d2, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, soff, 0));
    return {v.x, v.y};
}
__device__ __forceinline__ void res_store16(res_rsrc r, uint32_t lane_off, uint32_t soff, double x, double y) {
    const res_u4 bits = __builtin_bit_cast(res_u4, res_d2{x, y});
    __builtin_amdgcn_raw_buffer_store_b128(bits, r, lane_off, soff, 0);
    // the data registers stay untouched for two more issue slots: the compiler pads the "store of more than 8 bytes, then a write
    // of its data registers" hazard only when the scalar offset is an immediate (round 5, tools/exp/stft_pk16r.h)
    asm volatile("s_nop 1" ::"v"(bits));
}

// A value the compiler must treat as produced HERE: table loads addressed through it cannot be hoisted above this point (they are
// loads of read-only memory: without this every quad's table entries are fetched at the top of the kernel and spilled), and
// `chain` orders the point behind the arithmetic that produced it (one quad's loads at a time).
__device__ __forceinline__ int res_opaque(int v, double& chain) {
    asm volatile("" : "+v"(v), "+v"(chain));
    return v;
}

// The 6000-point transform, 6 x 10 x 10 x 10 as fft_static.h has it (same tables), with the powers of a butterfly's twiddle formed as a
// chain w, w w1, ... and applied as they are formed: two factors alive instead of the ten of the log-depth scheme — the transform
// runs in 106 / 74 registers (512 / 768 threads) instead of 120 / 80, which is what the parked sub-spectra leave it.
// (Round 6, sessions r6i-r6l, all measured equal within 1-3 % and not kept: signal 1's samples requested before signal 0's second
// transform — also with the transform's tables in LDS and no spill reload in between, so that nothing waits for vector memory while
// they are in flight: the phase that issues them grows by what the load phase shrinks (the requests queue at the chip's memory, the
// issuing wave with them); the first round of workgroups started in eight groups 0.8-7 us apart: 1024 pairs ±2 %, 4096 pairs −4 %.
// A pair costs a CU 83-88 us with the chip full whether or not its neighbours are in the same phase, 60 alone.)
// (Signal 1's 24 sample loads requested before signal 0's second transform — 96 registers in flight across it, 16 spilled — measure
// equal too: 100 / 256 / 1024 pairs 75.4 / 90.7 / 358 us against 74.4 / 86.9 / 349, session r6i: with every CU in the same phase the load
// phase is the chip's memory stream, not a latency.)
// (Requesting a pass's table entry one pass ahead — a pass opens with a wait for it — costs 8 registers = 16 spilled, and measures equal:
// 100 / 256 / 1024 pairs 75.0 / 89.5 / 357 us against 74.7 / 87.3 / 358, session r6h.)
template <int NT, int R, int P>
__device__ __forceinline__ void res_pass(cpx<double>* buf, const cpx<double>* __restrict__ tw, int tid) {
    using C = cpx<double>;
    constexpr int NB = kResM2 / R;                     // butterflies
    constexpr int B = (NB + NT - 1) / NT;              // per thread
    C v[B][R];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = tid + b * NT;
        if (B * NT == NB || j < NB) {
#pragma unroll
            for (int q = 0; q < R; ++q) v[b][q] = buf[j + q * NB];
            if constexpr (P > 1) {
                const C w1 = tw[j % P];
                C w = w1;
                v[b][1] = cmul(v[b][1], w);
#pragma unroll
                for (int q = 2; q < R; ++q) {
                    w = cmul(w, w1);
                    v[b][q] = cmul(v[b][q], w);
                }
            }
            dft_static<double, R>(v[b]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int j = tid + b * NT;
        if (B * NT == NB || j < NB) {
            const int k = P > 1 ? j % P : 0;
            const int base = (j - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; ++q) buf[base + q * P] = v[b][q];
        }
    }
    __syncthreads();
}
template <int NT>
__device__ __forceinline__ void res_fft(cpx<double>* buf, const cpx<double>* __restrict__ tws, int tid) {
    res_pass<NT, 6, 1>(buf, tws, tid);
    res_pass<NT, 10, 6>(buf, tws, tid);               // tables of the passes concatenated (make_static_twiddles): 6, 60, 600 entries
    res_pass<NT, 10, 60>(buf, tws + 6, tid);
    res_pass<NT, 10, 600>(buf, tws + 66, tid);
}

// The last pass of an INVERSE transform without its scatter: butterfly j leaves the points j + 600 q (q < 10) in the thread's registers
// — consecutive threads, consecutive points — and the correlation's rows are stored from there: no write of the pass's output into the
// transform array, no read of it back, one barrier fewer.  (A forward transform's output is wanted by quads: it goes through LDS.)
template <int NT>
struct ResLast {
    static constexpr int R = 10, PP = 600, NB = kResM2 / R;        // 600 butterflies
    static constexpr int B = (NB + NT - 1) / NT;
    static constexpr bool full(int b) { return (b + 1) * NT <= NB; }
};
template <int NT>
__device__ __forceinline__ void res_last_pass(const cpx<double>* buf, const cpx<double>* __restrict__ tw, int tid,
                                              cpx<double> (&v)[ResLast<NT>::B][10]) {
    using C = cpx<double>;
    using LP = ResLast<NT>;
#pragma unroll
    for (int b = 0; b < LP::B; ++b) {
        const int j = tid + b * NT;
#pragma unroll
        for (int q = 0; q < 10; ++q) v[b][q] = C{0.0, 0.0};
        if (LP::full(b) || j < LP::NB) {
#pragma unroll
            for (int q = 0; q < 10; ++q) v[b][q] = buf[j + q * LP::NB];
            const C w1 = tw[j];                                  // j mod 600 = j
            C w = w1;
            v[b][1] = cmul(v[b][1], w);
#pragma unroll
            for (int q = 2; q < 10; ++q) {
                w = cmul(w, w1);
                v[b][q] = cmul(v[b][q], w);
            }
            dft_static<double, 10>(v[b]);
        }
    }
}
template <int NT>
__device__ __forceinline__ void res_fft_head(cpx<double>* buf, const cpx<double>* __restrict__ tws, int tid) {      // passes 1 .. 3
    res_pass<NT, 6, 1>(buf, tws, tid);
    res_pass<NT, 10, 6>(buf, tws, tid);
    res_pass<NT, 10, 60>(buf, tws + 6, tid);
}

// Passes 2 .. 4 (the first pass was taken where the transform's input was formed: res_load_signal / res_second_half)
template <int NT>
__device__ __forceinline__ void res_fft_tail(cpx<double>* buf, const cpx<double>* __restrict__ tws, int tid) {
    res_pass<NT, 10, 6>(buf, tws, tid);
    res_pass<NT, 10, 60>(buf, tws + 6, tid);
    res_pass<NT, 10, 600>(buf, tws + 66, tid);
}

// The first pass of a forward transform has no twiddles and its butterflies are a thread's own to choose: butterfly j' takes the
// points j' + 1000 q (q < 6) — so a thread LOADS those points (consecutive threads, consecutive points: whole lines all the same),
// takes the 6-point DFT in registers and writes the pass's output 6 j' + q.  No write of the input into the transform array, no gather
// of it back, one barrier pair fewer per forward transform.  Slot s = 6 b + q of a thread: point (tid + NT b) + 1000 q.
template <int NT>
struct ResFirst {
    static constexpr int NB = kResM2 / 6;                           // 1000 butterflies
    static constexpr int B = (NB + NT - 1) / NT;                    // per thread
    static_assert(6 * B == ResPlan<NT>::NS, "a thread's first-pass inputs are its NS slots");
    static constexpr bool full(int b) { return (b + 1) * NT <= NB; }
};

// One signal, read once: (x w) of the even sample pairs into the transform array, of the odd ones into the thread's slots of the
// second array (the last slots: registers).  Returns the sum of the thread's samples (the mean leaves in the spectrum: rfft((x - m) w)
// = rfft(x w) - m rfft(w)).  Two batches of slots: 2 x 2 x NS / 2 loads of 16 bytes in flight per thread.
template <int NT>
__device__ __forceinline__ double res_load_signal(const double* __restrict__ sig, res_rsrc wrs, cpx<double>* buf, cpx<double>* sp,
                                                  cpx<double> (&Y)[ResPlan<NT>::NS - ResPlan<NT>::SPS], int tid) {
    using C = cpx<double>;
    using P = ResPlan<NT>;
    using F = ResFirst<NT>;
    const res_rsrc srs = res_make_rsrc(sig);
    uint32_t lane = (uint32_t)tid * 32u;                                            // point m of the signal: bytes 32 m ...
    asm volatile("" : "+v"(lane));      // (or the second signal re-uses the first one's window values: 2 NS x 4 registers kept from here to there)
    uint32_t lane_hi = lane + 16u;
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < F::B; ++b) {                                                // a batch = one first-pass butterfly: 4 x 6 loads in flight
        C x0[6], x1[6], w0[6], w1[6];
        const int jb = tid + b * NT;
        const bool live = F::full(b) || jb < F::NB;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const uint32_t soff = (uint32_t)(b * NT + 1000 * q) * 32u;
            x0[q] = x1[q] = w0[q] = w1[q] = C{0.0, 0.0};
            if (live) {
                x0[q] = res_load16(srs, lane, soff);
                x1[q] = res_load16(srs, lane_hi, soff);
                w0[q] = res_load16(wrs, lane, soff);
                w1[q] = res_load16(wrs, lane_hi, soff);
            }
        }
        C v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int slot = 6 * b + q;
            acc += (x0[q].x + x0[q].y) + (x1[q].x + x1[q].y);
            v[q] = {x0[q].x * w0[q].x, x0[q].y * w0[q].y};
            const C y1 = {x1[q].x * w1[q].x, x1[q].y * w1[q].y};
            if (slot < P::SPS) sp[slot * NT + tid] = y1;             // (a thread's slots of the second array are its own: no barrier)
            else Y[slot - P::SPS] = y1;
        }
        dft_static<double, 6>(v);
        if (live) {
#pragma unroll
            for (int q = 0; q < 6; ++q) buf[6 * jb + q] = v[q];
        }
        asm volatile("" : "+v"(lane), "+v"(lane_hi), "+v"(acc));   // the second batch's loads behind the first batch's arithmetic (the
                                                                   // scheduler otherwise issues all 4 NS loads first and spills them)
    }
    return acc;
}

// The waiting r = 1 half: its first pass straight from the thread's slots into the transform array (the array's readers are behind a
// barrier)
template <int NT>
__device__ __forceinline__ void res_second_half(cpx<double>* buf, const cpx<double>* sp, const cpx<double> (&Y)[ResPlan<NT>::NS - ResPlan<NT>::SPS],
                                                int tid) {
    using C = cpx<double>;
    using P = ResPlan<NT>;
    using F = ResFirst<NT>;
#pragma unroll
    for (int b = 0; b < F::B; ++b) {
        const int jb = tid + b * NT;
        C v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int slot = 6 * b + q;
            v[q] = slot < P::SPS ? sp[slot * NT + tid] : Y[slot < P::SPS ? 0 : slot - P::SPS];
        }
        dft_static<double, 6>(v);
        if (F::full(b) || jb < F::NB) {
#pragma unroll
            for (int q = 0; q < 6; ++q) buf[6 * jb + q] = v[q];
        }
    }
}

// The thread's elements of the sub-spectrum in the transform array: E[2 i], E[2 i + 1] = elements q, M2 - q of quad i
template <int NT>
__device__ __forceinline__ void res_take_quads(const cpx<double>* buf, cpx<double> (&E)[ResPlan<NT>::NE], int tid) {
    using P = ResPlan<NT>;
    asm volatile("" : "+v"(tid));       // (the mirrored elements' addresses re-made per call: kept across the kernel one of them was the
                                        // kernel's last spilled register — a scratch allocation for 4 bytes per lane)
#pragma unroll
    for (int i = 0; i < P::NQ; ++i) {
        const int q = tid + i * NT;
        E[2 * i] = E[2 * i + 1] = {0.0, 0.0};
        if (P::quad_full(i) || q <= kResM2 / 2) {
            E[2 * i] = buf[q];
            E[2 * i + 1] = buf[q == 0 ? 0 : kResM2 - q];
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(NT) gcc_phat_resident_kernel(const GccArgs a) {
    using C = cpx<double>;
    using P = ResPlan<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    C* buf = (C*)smem;                                  // the transform array, M2 points
    C* sp = buf + kResM2;                               // SPS slots per thread, element (slot, tid) at slot * NT + tid: thread-private
    double* red = (double*)(sp + P::SPS * NT);          // 2 x 16 doubles + 16 ints
    int* redi = (int*)(red + 32);
    const int tid = threadIdx.x;
    constexpr int M = kResM, M2 = kResM2, L = kResL, NQ = P::NQ, NS = P::NS, NE = P::NE, SPS = P::SPS;
    const res_rsrc wrs = res_make_rsrc(a.window), twm_rs = res_make_rsrc(a.twm), twl_rs = res_make_rsrc(a.twl), dw_rs = res_make_rsrc(a.dw);
    const C* __restrict__ tws = (const C*)a.tws;

    // (one pair per workgroup: as a loop over pairs the compiler hoists the loop-invariant table loads of the cross spectrum out of it
    // and spills them)
    const int pair = blockIdx.x;
    const double* const sig0 = a.d0 + (size_t)pair * L;
    const double* const sig1 = a.d1 + (size_t)pair * L;
    GCC_STAMP(0);
    // ---- signal 0: both sub-spectra into registers ----------------------------------------------------------------------
    C Y[NS - SPS];
    C e00[NE], e01[NE];
    // (the wavefronts' sample sums wait in LDS for the cross spectrum, not in registers across four transforms)
    auto leave_sum = [&](double v, int at) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) red[at + (tid >> 6)] = v;
    };
    leave_sum(res_load_signal<NT>(sig0, wrs, buf, sp, Y, tid), 0);
    __syncthreads();
    GCC_STAMP(1);
    res_fft_tail<NT>(buf, tws, tid);
    GCC_STAMP(2);
    res_take_quads<NT>(buf, e00, tid);
    __syncthreads();
    res_second_half<NT>(buf, sp, Y, tid);
    __syncthreads();
    res_fft_tail<NT>(buf, tws, tid);
    res_take_quads<NT>(buf, e01, tid);
    __syncthreads();
    GCC_STAMP(3);
    // ---- signal 1: first sub-spectrum into the second LDS array (the thread's first SPS elements) and registers (the others) --
    leave_sum(res_load_signal<NT>(sig1, wrs, buf, sp, Y, tid), 16);
    __syncthreads();
    res_fft_tail<NT>(buf, tws, tid);
    C e10r[NE - SPS];
    {
        C te[NE];
        res_take_quads<NT>(buf, te, tid);
        __syncthreads();
        res_second_half<NT>(buf, sp, Y, tid);                // (reads the thread's slots of the second array ...
#pragma unroll
        for (int e = 0; e < NE; ++e) {                       //  ... before these writes re-use them: same thread, program order)
            if (e < SPS) sp[e * NT + tid] = te[e];
            else e10r[e - SPS] = te[e];
        }
    }
    __syncthreads();
    res_fft_tail<NT>(buf, tws, tid);
    GCC_STAMP(4);
    // ---- the means ---------------------------------------------------------------------------------------------------------
    double mean0 = 0.0, mean1 = 0.0;                     // (the sums were left before barriers long past)
#pragma unroll
    for (int w = 0; w < P::NW; ++w) {
        mean0 += red[w];
        mean1 += red[16 + w];
    }
    mean0 /= (double)L;
    mean1 /= (double)L;
    if (tid == 0 && a.means) {
        a.means[2 * pair] = mean0;
        a.means[2 * pair + 1] = mean1;
    }
    // ---- cross spectrum of the thread's quads; G replaces signal 0's elements in their registers -------------------------
    // g[i][0 .. 3] = G at the bins q, M - q, M2 - q, M2 + q
    C g[NQ][4];
    double gmax2 = 0.0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * NT;
        const bool valid = P::quad_full(i) || q <= M2 / 2;
        const int qc = res_opaque(valid ? q : 0, gmax2);
        const int eb = qc == 0 ? 0 : M2 - qc;
        const uint32_t oq = (uint32_t)qc * 16u, om = (uint32_t)(M2 - qc) * 16u;
        const C t = res_load16(twm_rs, oq, 0), tk = res_load16(twl_rs, oq, 0);
        const C dwq[4] = {res_load16(dw_rs, oq, 0), res_load16(dw_rs, om, M2 * 16u), res_load16(dw_rs, om, 0), res_load16(dw_rs, oq, M2 * 16u)};
        C a1, c1;
        if (2 * i < SPS) a1 = sp[(2 * i) * NT + tid];
        else a1 = e10r[2 * i < SPS ? 0 : 2 * i - SPS];
        if (2 * i + 1 < SPS) c1 = sp[(2 * i + 1) * NT + tid];
        else c1 = e10r[2 * i + 1 < SPS ? 0 : 2 * i + 1 - SPS];
        const C b1 = buf[qc], d1 = buf[eb];
        const C tkm = {-tk.x, tk.y};                     // twl[M - q]  = -conj(twl[q])
        const C tk2m = {-tk.y, -tk.x};                   // twl[M2 - q] = -i conj(twl[q])
        const C tk2p = {tk.y, -tk.x};                    // twl[M2 + q] = -i twl[q]
        C D[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const C sa = s == 0 ? e00[2 * i] : a1, sb = s == 0 ? e01[2 * i] : b1, sc = s == 0 ? e00[2 * i + 1] : c1,
                    sd = s == 0 ? e01[2 * i + 1] : d1;
            const double mean = s == 0 ? mean0 : mean1;
            const C tb = cmul(t, sb), td = cmul(cconj(t), sd);
            const C Zq = sa + tb, Z2p = sa - tb, ZMq = sc + td, Z2m = sc - td;
            D[s][0] = gcc_unpack(Zq, ZMq, tk);
            D[s][1] = gcc_unpack(ZMq, Zq, tkm);
            D[s][2] = gcc_unpack(Z2m, Z2p, tk2m);
            D[s][3] = gcc_unpack(Z2p, Z2m, tk2p);
#pragma unroll
            for (int n = 0; n < 4; ++n) D[s][n] = {D[s][n].x - mean * dwq[n].x, D[s][n].y - mean * dwq[n].y};
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            g[i][n] = valid ? cmul(cconj(D[0][n]), D[1][n]) : C{0.0, 0.0};
            gmax2 = fmax(gmax2, g[i][n].x * g[i][n].x + g[i][n].y * g[i][n].y);
        }
    }
    // block maximum (its barriers: every read of the transform array is done)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax2 = fmax(gmax2, __shfl_down(gmax2, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = gmax2;
    __syncthreads();
    gmax2 = red[0];
#pragma unroll
    for (int w = 1; w < P::NW; ++w) gmax2 = fmax(gmax2, red[w]);
    const double gmax = sqrt(gmax2);
    GCC_STAMP(5);
    // ---- PHAT weights, Hermitian packing, the radix-2 step of the inverse: all inside the quad --------------------------
    // (zk, zm) = Zi[k], Zi[M - k] from the weighted cross spectrum at (k, M - k); tk, tm = conj(twl[k]), conj(twl[M - k])
    // The PHAT weight 1 / (1e-10 max|G| + |G|) from the squared magnitude: v_rsq_f64 / v_rcp_f64 seeds (~23 bits) with two Newton steps
    // each (full float64 precision: the weights differ from sqrt + division by a few 1e-16, the correlation's bar is 1e-9) — ~20
    // instructions instead of the ~60 of the correctly rounded sqrt and quotient, 24 weights per thread: the phase was 4.5 us of a pair.
    const double wfloor = 1e-10 * gmax;
    auto phat_weight = [&](double n2) -> double {
        const double x = fmax(n2, 1e-290);                // (|G| = 0: sqrt(x) = 1e-145, nothing beside the floor)
        double r = __builtin_amdgcn_rsq(x);
        r = r * __builtin_fma(-0.5 * x, r * r, 1.5);
        r = r * __builtin_fma(-0.5 * x, r * r, 1.5);
        const double d = __builtin_fma(x, r, wfloor);     // floor + sqrt(x)
        double q = __builtin_amdgcn_rcp(d);
        q = q * __builtin_fma(-d, q, 2.0);
        q = q * __builtin_fma(-d, q, 2.0);
        return q;
    };
    auto pack = [&](C A, C B, C tk, C tm, bool edge, C& zk, C& zm) {
        const double wa = phat_weight(A.x * A.x + A.y * A.y);
        const double wb = phat_weight(B.x * B.x + B.y * B.y);
        A = {A.x * wa, A.y * wa};
        B = {B.x * wb, B.y * wb};
        if (edge) A.y = B.y = 0.0;                       // irfft ignores the imaginary part of the edge bins 0 and M
        {
            const C Bc = cconj(B), Sm = A + Bc, Dd = A - Bc, u = cmul(tk, Dd);
            zk = {0.5 * (Sm.x - u.y), 0.5 * (Sm.y + u.x)};
        }
        {
            const C Ac = cconj(A), Sm = B + Ac, Dd = B - Ac, u = cmul(tm, Dd);
            zm = {0.5 * (Sm.x - u.y), 0.5 * (Sm.y + u.x)};
        }
    };
    C x1[NE];                                            // the r = 1 inverse input at q and M2 - q, conjugated
    double chain = gmax;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * NT;
        const bool valid = P::quad_full(i) || q <= M2 / 2;
        const uint32_t oq = (uint32_t)res_opaque(valid ? q : 0, chain) * 16u;
        const C t = res_load16(twm_rs, oq, 0), tk = res_load16(twl_rs, oq, 0);
        C zq, zMq, z2m, z2p;
        // conj(twl[q]), conj(twl[M - q]) = -twl[q]
        pack(g[i][0], g[i][1], cconj(tk), C{-tk.x, -tk.y}, q == 0, zq, zMq);
        // conj(twl[M2 - q]) = i twl[q], conj(twl[M2 + q]) = i conj(twl[q])
        pack(g[i][2], g[i][3], C{-tk.y, tk.x}, C{tk.y, tk.x}, false, z2m, z2p);
        // element q: Zi[q] +- Zi[M2 + q]; element M2 - q: Zi[M2 - q] +- Zi[M - q]; W_M^{-(M2 - q)} = -twm[q]
        const C s0q = zq + z2p, s1q = cmul(cconj(t), zq - z2p);
        const C s0m = z2m + zMq, s1m = cmul(C{-t.x, -t.y}, z2m - zMq);
        x1[2 * i] = cconj(s1q);
        x1[2 * i + 1] = cconj(s1m);
        chain = x1[2 * i + 1].y;                         // (the next quad's table loads wait for this quad's arithmetic)
        if (valid) {
            buf[q] = cconj(s0q);
            if (q > 0) buf[M2 - q] = cconj(s0m);
        }
    }
    __syncthreads();
    GCC_STAMP(6);
    // ---- inverse sub-transforms: z[2 m + r] = (1/M) conj(FFT_M2(conj input_r))[m] ---------------------------------------
    // The last pass leaves butterfly j's points m = j + 600 q in registers (res_last_pass): the r = 0 half waits there for the r = 1
    // half, and a thread stores the 32 contiguous bytes of every m it holds.
    using LP = ResLast<NT>;
    res_fft_head<NT>(buf, tws, tid);
    const double inv = 1.0 / (double)M;
    C o0[LP::B][10];
    res_last_pass<NT>(buf, tws + 66, tid, o0);
    __syncthreads();                                     // every read of the transform array is done
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * NT;
        if (P::quad_full(i) || q <= M2 / 2) {
            buf[q] = x1[2 * i];
            if (q > 0) buf[M2 - q] = x1[2 * i + 1];
        }
    }
    __syncthreads();
    res_fft_head<NT>(buf, tws, tid);
    C o1[LP::B][10];
    res_last_pass<NT>(buf, tws + 66, tid, o1);
    GCC_STAMP(7);
    const res_rsrc out_rs = res_make_rsrc(a.xcorr + (size_t)pair * L);
    double best = -1.0;
    int besti = 0;
#pragma unroll
    for (int b = 0; b < LP::B; ++b) {
        const int jb = tid + b * NT;
        if (LP::full(b) || jb < LP::NB) {
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int t = 4 * (jb + LP::NB * q);            // samples t .. t + 3 = (z[2 m], z[2 m + 1]), m = j + 600 q
                const uint32_t soff = (uint32_t)(b * NT + LP::NB * q) * 32u;
                const double r0 = o0[b][q].x * inv, i0 = -o0[b][q].y * inv, r1 = o1[b][q].x * inv, i1 = -o1[b][q].y * inv;
                res_store16(out_rs, (uint32_t)tid * 32u, soff, r0, i0);
                res_store16(out_rs, (uint32_t)tid * 32u + 16u, soff, r1, i1);
                // (a thread's points do not come in ascending order: ties go to the smaller index explicitly)
                if (fabs(r0) > best || (fabs(r0) == best && t < besti)) { best = fabs(r0); besti = t; }
                if (fabs(i0) > best || (fabs(i0) == best && t + 1 < besti)) { best = fabs(i0); besti = t + 1; }
                if (fabs(r1) > best || (fabs(r1) == best && t + 2 < besti)) { best = fabs(r1); besti = t + 2; }
                if (fabs(i1) > best || (fabs(i1) == best && t + 3 < besti)) { best = fabs(i1); besti = t + 3; }
            }
        }
    }
    // ---- argmax |xcorr| (first index on ties, as numpy.argmax) -----------------------------------------------------------
    if (a.argmax) {
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_down(best, o, 64);
            const int oi = __shfl_down(besti, o, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < P::NW; ++w)
                if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
            a.argmax[pair] = besti;
        }
    }
    GCC_STAMP(8);
}

}  // namespace frt
