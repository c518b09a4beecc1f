"""Per-kernel measurements on one MI355X with the CPU oracle timed beside each (1 host core).

    python tools/bench_all.py > gpurun_out/bench_all.json

One JSON object per line: kernel, workload, units/s on the GPU, algorithmic GB/s (SURVEY.md §8d
byte counts), the oracle's units/s on a bounded sample, and the parity error measured on that sample.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def timeit(fn, sync, iters, prewarm_s=0.25):
    """Average time of fn over `iters` back-to-back calls, after `prewarm_s` of the same calls: after idle the
    GPU needs tens of milliseconds of continuous work to reach its sustained clocks."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm_s:
        fn()
        sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    sync()
    return (time.perf_counter() - t0) / iters


def main():
    import torch

    from friture_amd import _lib, filter_design, palette, tables
    from friture_amd.filter import IirBank
    from friture_amd.octavefilters import Octave_Filters
    from friture_amd.signal.correlation import GccPhat
    from friture_amd.stft import StftEngine
    from oracle import dsp
    _lib.init(0)
    sync = torch.cuda.synchronize
    rng = np.random.default_rng(42)
    t = filter_design.load_tables()
    out = []

    # ---- K1 at the BASELINE sizes -----------------------------------------------------------------
    for n_fft, hop, ch, log2t, kind, name in [(1024, 512, 1, 26, 3, "configs[1] spectrogram image"),
                                              (1024, 256, 1, 26, 0, "reference default overlap 75 %, PSD"),
                                              (16384, 8192, 32, 20, 0, "configs[3] shard of 32 ch, PSD"),
                                              (16384, 4096, 32, 20, 0, "configs[3] shard, 75 % overlap, PSD"),
                                              (4096, 1024, 16, 22, 3, "spectrogram default N=4096, image")]:
        T = 1 << log2t
        # three distinct batches, visited in turn: every launch reads its samples from HBM (a single batch
        # re-processed every launch partly survives in the 256 MB Infinity Cache; that rate is `gpu_same_batch`)
        xb = [torch.from_numpy((0.25 * rng.standard_normal((ch, T))).astype(np.float32)).cuda() for _ in range(3)]
        x = xb[0]
        eng = StftEngine(n_fft, hop, ch, 32)
        eng.set_epilogue(tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0], -140.0, 0.0, palette.cmr_lut())
        F = eng.frames_for(T)
        ob = [torch.empty((ch, F, n_fft // 2 + 1), dtype=torch.int32 if kind == 3 else torch.float32, device="cuda") for _ in range(3)]
        o = ob[0]
        turn = [0]

        def rotating():
            b = turn[0] % 3
            turn[0] += 1
            eng.run(kind, xb[b], ob[b])

        dt = timeit(rotating, sync, 30)
        dt_same = timeit(lambda: eng.run(kind, x, o), sync, 30)
        nb = 4 * hop + 4 * (n_fft // 2 + 1)
        xs = x[0, : n_fft + hop * 255].cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        ref = dsp.stft_psd(xs, n_fft, hop)
        cpu = 256 / (time.perf_counter() - t0)
        got = StftEngine(n_fft, hop, 1, 32).psd(xs.astype(np.float32)[None, :])[0]
        err = float(np.max(np.max(np.abs(got - ref), axis=1) / np.max(ref, axis=1)))
        out.append(dict(kernel="K1 stft_kernel", workload=f"{name}: N={n_fft} hop={hop} C={ch} T=2^{log2t}", unit="spectra/s",
                        gpu=ch * F / dt, ms=dt * 1e3, algorithmic_GBps=ch * F * nb / dt / 1e9, frac_of_8TBps=ch * F * nb / dt / 8e12,
                        cpu_oracle=cpu, parity_rel_max=err, gpu_same_batch=ch * F / dt_same))
        del x, o, xb, ob

    # ---- K2/K4 exact IIR bank energies ---------------------------------------------------------------
    for ch, bpo, log2n, chunk in [(8, 3, 22, 2048), (64, 24, 20, 8192), (8, 3, 16, 0)]:
        n = 1 << log2n
        bank = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
        bank.set_chunk(chunk)
        x = torch.from_numpy((0.25 * rng.standard_normal((ch, n))).astype(np.float32)).cuda()
        alphas, kernels = dsp.band_smoothing_setup(bpo, 1.0)
        o = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device="cuda")
        dt = timeit(lambda: bank.energies(x, 1024, np.array(alphas), out=o), sync, 3)
        units = ch * (n // 1024) * 9 * bpo
        # oracle on 64 blocks of channel 0 (C recurrence + numpy), parity of the energies
        blocks = 64
        xs = x[0, : 1024 * blocks].cpu().numpy().astype(np.float64)
        bank1 = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), 1)
        got = bank1.energies(xs.astype(np.float32)[None, :], 1024, np.array(alphas))[0]
        boct, aoct = list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"])
        zs = dsp.iir_bank_filtic(t["bdec"], t["adec"], boct, aoct)
        prev = [0.0] * (9 * bpo)
        t0 = time.perf_counter()
        worst = 0.0
        for b in range(blocks):
            y, _, zs = dsp.iir_bank(t["bdec"], t["adec"], boct, aoct, xs[b * 1024:(b + 1) * 1024], zs)
            prev = dsp.band_energies(y, kernels, alphas, prev)
            worst = max(worst, float(np.max(np.abs(got[b] / np.array(prev) - 1))))
        cpu = blocks * 9 * bpo / (time.perf_counter() - t0)
        out.append(dict(kernel="K2/K4 iir_stage_kernel + energy_scan_kernel",
                        workload=f"exact IIR bank energies: C={ch} bpo={bpo} T=2^{log2n} chunk={chunk}", unit="octave-bands/s",
                        gpu=units / dt, ms=dt * 1e3, samples_per_s=ch * n / dt,
                        algorithmic_GBps=ch * (n // 1024) * (4096 + 4 * 9 * bpo) / dt / 1e9, cpu_oracle=cpu, parity_rel_max=worst))
        del x, o

    # ---- K3 FFT overlap-add bank (streaming, one 1024-sample block per call, host buffers) ------------
    for bpo in (3, 24):
        of = Octave_Filters(bpo)
        ref = dsp.OlaBank(bpo)
        blk = (0.25 * rng.standard_normal(1024))
        dt = timeit(lambda: of.filter(blk), lambda: None, 50, prewarm_s=0.05)
        t0 = time.perf_counter()
        for _ in range(20):
            yr, _ = ref.filter(blk)
        cpu = 20 * 9 * bpo / (time.perf_counter() - t0)
        of.reset()
        ref.reset()
        y, _ = of.filter(blk)
        yr, _ = ref.filter(blk)
        err = max(float(np.max(np.abs(a - b)) / np.max(np.abs(b))) for a, b in zip(y, yr))
        out.append(dict(kernel="K3 ola_stage_kernel", workload=f"Octave_Filters({bpo}).filter, 1024-sample block, host buffers (PCIe + 9 launches)",
                        unit="octave-bands/s", gpu=9 * bpo / dt, ms=dt * 1e3, cpu_oracle=cpu, parity_rel_max=err))

    # ---- K5 GCC-PHAT --------------------------------------------------------------------------------------
    for L, pairs in [(24000, 100), (24000, 1024), (24000, 1)]:
        d0 = 0.25 * rng.standard_normal((pairs, L))
        d1 = np.roll(d0, 37, axis=1) + 0.025 * rng.standard_normal((pairs, L))
        g = GccPhat(L, pairs)
        a0, a1 = torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()
        dt = timeit(lambda: g.correlate(a0, a1), sync, 10)
        t0 = time.perf_counter()
        ref, _, _ = dsp.gcc_phat(d0[0], d1[0])
        cpu = 1 / (time.perf_counter() - t0)
        xg, am = g.correlate(a0, a1)
        err = float(np.max(np.abs(xg[0].cpu().numpy() - ref)) / np.max(np.abs(ref)))
        out.append(dict(kernel="K5 gcc_phat_kernel", workload=f"GCC-PHAT L={L}, {pairs} window pair(s), device resident f64", unit="windows/s",
                        gpu=pairs / dt, ms=dt * 1e3, algorithmic_GBps=pairs * 24 * L / dt / 1e9, cpu_oracle=cpu, parity_rel_max=err,
                        argmax=int(am[0])))
    # ---- T1 pitch tracker ---------------------------------------------------------------------------------
    from friture_amd.pitch_tracker import PitchEngine, swipe_tables
    grid, _, kern = swipe_tables()
    for n_fft, hop, ch, log2t in [(4096, 1024, 8, 22), (1024, 256, 8, 22)]:
        T = 1 << log2t
        tt = np.arange(T)
        f_path = 110.0 * 2 ** (2.0 * tt / T)
        ph = 2 * np.pi * np.cumsum(f_path) / 48000.0
        base = 0.2 * (np.sin(ph) + 0.6 * np.sin(2 * ph) + 0.3 * np.sin(3 * ph))
        x = torch.from_numpy(np.stack([base + 1e-3 * rng.standard_normal(T) for _ in range(ch)])).cuda()
        eng = PitchEngine(n_fft, hop, ch, grid=grid, kernels=kern)
        F = eng.frames_for(T)
        dt = timeit(lambda: eng.track(x), sync, 5)
        xs = x[0, : n_fft + hop * 63].cpu().numpy()
        t0 = time.perf_counter()
        ref = dsp.pitch_track(xs, n_fft, hop, grid, kern)
        cpu = 64 / (time.perf_counter() - t0)
        got = PitchEngine(n_fft, hop, 1, grid=grid, kernels=kern).track(xs)[0]
        ok = np.array_equal(np.isnan(got), np.isnan(ref[0]))
        m = ~np.isnan(ref[0])
        err = float(np.max(np.abs(got[m] / ref[0][m] - 1))) if ok and m.any() else float("nan")
        flops = 2.0 * kern.shape[0] * kern.shape[1]
        out.append(dict(kernel="T1 pitch_strength_kernel (+ K1 f64, loggrid, pick, gate)",
                        workload=f"pitch tracker N={n_fft} hop={hop} C={ch} T=2^{log2t}, 481 candidates x 1023 grid points, f64",
                        unit="frames/s", gpu=ch * F / dt, ms=dt * 1e3, contraction_TFLOPs=ch * F * flops / dt / 1e12,
                        cpu_oracle=cpu, parity_rel_max=err, voiced_pattern_equal=bool(ok)))
        del x
    for rec in out:
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
