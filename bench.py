#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): the rolling
spectrogram of one 48 kHz channel per GPU — 1024-point Hann STFT at 50 % overlap, dB,
A-weighting, normalisation to [-140, 0] dB and the colour look-up — over T = 2^26 synthetic
samples resident in HBM (131 071 spectra per channel per step).  A step is one pass of the hot
path over one such batch; the steps rotate over --batches distinct batches (default 3, different
seeds, separate output buffers) so that every step reads its samples from HBM — with a single
268 MB batch re-processed every step, part of it can survive in the 256 MB Infinity Cache between
steps (it does when the rows are stored non-temporally; reported separately as `same_batch`).
Before anything is timed the GPU is brought to its sustained clocks with --prewarm-ms of the same
launches: straight after start-up a 55-launch run sits inside the clock ramp and reads 17 % slow.  With N GPUs every rank owns its own channel(s) (weak scaling, no data-path
collective); `value` is the whole-job spectra/s: total spectra of all ranks / max-over-ranks time.

One JSON line is printed by rank 0; besides the contract fields it carries
  roofline      HBM roofline of the dominant kernel (stft_kernel): algorithmic bytes per launch
                (4*hop + 4*(N/2+1) = 4100 B per spectrum) / average launch duration measured with
                HIP events on the launch stream, against the 8 TB/s peak
  cpu_baseline  the oracle (numpy restatement of the reference path, 1 core) timed on a bounded
                sample of the same input on this box's host
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)


def synth_channel(channel: int, n: int) -> np.ndarray:
    """S-noise of SURVEY.md §8d: 0.25 * standard_normal, seed 42 + channel, float32 PCM."""
    return (0.25 * np.random.default_rng(42 + channel).standard_normal(n, dtype=np.float32)).astype(np.float32)


def cpu_baseline(x: np.ndarray, n_fft: int, hop: int, weight, lut, budget_s: float):
    """Oracle timed on the host over a bounded sample of the same channel: on one core (about 60 % of
    budget_s seconds of CPU work: a prefix of the channel, repeated when the whole channel is shorter),
    then on every core at once (a pool of spawned workers, each running the same prefix)."""
    from oracle import cpu_bench, dsp
    frames_total = (len(x) - n_fft) // hop + 1
    probe = 2048
    t0 = time.perf_counter()
    dsp.spectrogram_image(x[: n_fft + hop * (probe - 1)].astype(np.float64), n_fft, hop, weight, -140.0, 0.0, lut)
    per = (time.perf_counter() - t0) / probe
    single_s = 0.6 * budget_s
    frames = int(min(frames_total, max(probe, single_s / per)))
    xs = x[: n_fft + hop * (frames - 1)].astype(np.float64)
    t0 = time.perf_counter()
    dsp.spectrogram_image(xs, n_fft, hop, weight, -140.0, 0.0, lut)
    first = time.perf_counter() - t0
    passes = 1 + max(0, int(round((single_s - first) / first)))
    for _ in range(passes - 1):
        dsp.spectrogram_image(xs, n_fft, hop, weight, -140.0, 0.0, lut)
    dt = time.perf_counter() - t0
    result = {"value": frames * passes / dt, "unit": "spectra/s", "cores": 1, "kind": "port",
              "sample": f"{passes} pass(es) over the first {frames} of {frames_total} spectra of channel 0 (same input), "
                        f"numpy float64 oracle of audioproc.analyzelive + dB + A-weighting + colour LUT, {dt:.1f} s",
              "host_cpus": os.cpu_count()}
    # every core: one spawned worker per core (workers import numpy + the oracle only, never the GPU runtime)
    try:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        cores = max(1, min(os.cpu_count() or 1, 64))
        wframes = int(min(frames, max(probe, 0.2 * budget_s / per)))
        job = (0, n_fft, hop, wframes, 1, np.asarray(weight, np.float64), np.asarray(lut, np.uint32))
        ctx = mp.get_context("spawn")
        with ProcessPoolExecutor(max_workers=cores, mp_context=ctx, initializer=cpu_bench.init_worker,
                                 initargs=(ctx.Barrier(cores),)) as pool:
            list(pool.map(cpu_bench.wait_ready, range(cores)))                  # every worker up and imported
            t0 = time.perf_counter()
            done = list(pool.map(cpu_bench.spectrogram_passes, [job] * cores))
            wall = time.perf_counter() - t0
        result["all_cores"] = {"value": sum(d[0] for d in done) / wall, "unit": "spectra/s", "cores": cores,
                               "sample": f"{cores} worker processes x {wframes} spectra of the same prefix, {wall:.1f} s"}
    except Exception as exc:                                                     # a reported extra, never fatal
        result["all_cores"] = {"error": repr(exc)}
    return result


def octave_band_leg(dev, world, rank, steps=10):
    """Second half of the BASELINE metric: octave-bands/s of the exact IIR 1/3-octave bank
    (BASELINE configs[2]: 8 ch per GPU, 48 kHz, 2^22 samples, band energies per 1024-sample block)."""
    import torch

    from friture_amd import distributed, filter_design
    from friture_amd.filter import IirBank
    t = filter_design.load_tables()
    ch, bpo, n = 8, 3, 1 << 22
    bank = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
    bank.set_chunk(2048)
    x = torch.from_numpy(np.stack([synth_channel(1000 + rank * ch + c, n) for c in range(ch)])).to(dev)
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])     # octavespectrum.py:145-153
    out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.25:                 # clock ramp, see main()
        bank.energies(x, 1024, alphas, out=out)
        torch.cuda.synchronize()
    distributed.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        bank.energies(x, 1024, alphas, out=out)
    torch.cuda.synchronize()
    distributed.barrier(dev)
    dt = distributed.max_over_ranks(time.perf_counter() - t0, dev) / steps
    units = world * ch * (n // 1024) * 9 * bpo
    return {"value": units / dt, "unit": "octave-bands/s", "ms_per_step": dt * 1e3,
            "config": f"exact IIR 1/3-octave bank (27 bands), {ch} ch/GPU x 2^22 samples, energies per 1024-sample block, "
                      f"time-parallel chunks of 2048", "algorithmic_GBps": world * ch * (n // 1024) * (4096 + 4 * 27) / dt / 1e9}


def pmc_traffic(n_fft: int, hop: int, frames: int):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this command, if any."""
    p = ROOT / "profiles" / "pmc_traffic.json"
    if not p.exists():
        return None
    try:
        rec = json.loads(p.read_text())
        if rec.get("n_fft") == n_fft and rec.get("hop") == hop and rec.get("frames") == frames:
            return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--fft-size", type=int, default=1024)
    ap.add_argument("--hop", type=int, default=0, help="default fft_size/2 (50 %% overlap)")
    ap.add_argument("--log2-samples", type=int, default=26)
    ap.add_argument("--channels-per-gpu", type=int, default=1)
    ap.add_argument("--kind", choices=["image", "psd", "db"], default="image")
    ap.add_argument("--batches", type=int, default=3, help="distinct input batches the steps rotate over (1 = same batch every step)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed launches before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    import torch

    from friture_amd import _lib, distributed, palette, tables
    from friture_amd.stft import StftEngine

    rank, local_rank, world = distributed.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.init(local_rank)

    n_fft = args.fft_size
    hop = args.hop or n_fft // 2
    T = 1 << args.log2_samples
    cpg = args.channels_per_gpu
    n_channels = cpg * world
    my_channels = distributed.shard_channels(n_channels, rank, world)

    # constant tables: computed on rank 0, broadcast over RCCL so every rank uses identical bits
    consts = {"weight": np.zeros(n_fft // 2 + 1), "lut": np.zeros(256, np.uint32)}
    if rank == 0:
        consts["weight"] = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
        consts["lut"] = palette.cmr_lut()
    consts = distributed.broadcast_tables(consts, src=0, device=dev)

    host_x = [synth_channel(c, T) for c in my_channels]
    nbatch = max(1, args.batches)
    xs = [torch.from_numpy(np.stack(host_x)).to(dev)]
    for b in range(1, nbatch):                                  # further batches: same statistics, other seeds
        xs.append(torch.from_numpy(np.stack([synth_channel(100000 * b + c, T) for c in my_channels])).to(dev))
    x = xs[0]
    eng = StftEngine(n_fft, hop, len(my_channels), 32)
    eng.set_epilogue(consts["weight"], -140.0, 0.0, consts["lut"])
    kind = {"image": 3, "psd": 0, "db": 1}[args.kind]
    F = eng.frames_for(T)
    outs = [torch.empty((len(my_channels), F, n_fft // 2 + 1), dtype=torch.int32 if kind == 3 else torch.float32, device=dev)
            for _ in range(nbatch)]
    out = outs[0]

    def timed(steps, rotate):
        """K launches bracketed by one HIP event pair on the launch stream; returns (wall s, kernel ms per launch)."""
        torch.cuda.synchronize()
        distributed.barrier(dev)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for k in range(steps):
            b = k % nbatch if rotate else 0
            eng.run(kind, xs[b], outs[b])         # one kernel launch on torch's current stream
        ev1.record()
        torch.cuda.synchronize()
        distributed.barrier(dev)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / steps

    # Clock ramp: after idle the GPU needs tens of milliseconds of continuous work before it runs at its sustained
    # clocks; 55 launches (8 ms) straight after start-up read ~17 % slow.  Pre-warm with the same launches for
    # --prewarm-ms (default 300 ms, untimed), then the contract's W warm-up steps, then the K timed steps.
    t_pre = time.perf_counter()
    k = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(64):
            eng.run(kind, xs[k % nbatch], outs[k % nbatch])
            k += 1
        torch.cuda.synchronize()
    for k in range(args.warmup):
        eng.run(kind, xs[k % nbatch], outs[k % nbatch])
    # One HIP event pair brackets the K launches on the launch stream (torch's current stream, which
    # the C ABI launches on): average launch duration = elapsed / K.  (An event pair *per launch*
    # would put two barrier packets between consecutive kernels and cost ~15 us per step.)
    wall, kernel_ms = timed(args.steps, rotate=True)
    elapsed = distributed.max_over_ranks(wall, dev)
    kernel_ms_max = distributed.max_over_ranks(kernel_ms, dev)
    same_batch_ms = None
    if nbatch > 1:
        same_batch_ms = distributed.max_over_ranks(timed(args.steps, rotate=False)[1], dev)
    # the same transform with its plain PSD output (no dB / weighting / colour epilogue), for reference
    psd_ms = None
    if kind == 3:
        image_outs, kind = outs, 0
        outs = [o.view(torch.float32) for o in image_outs]
        for k in range(args.warmup):
            eng.run(kind, xs[k % nbatch], outs[k % nbatch])
        psd_ms = distributed.max_over_ranks(timed(args.steps, rotate=True)[1], dev)
        outs, kind = image_outs, 3
        eng.run(kind, xs[0], outs[0])             # leave the image of batch 0 in place for the digest

    # post-batch summary gather (outside the timed region): per-channel mean pixel/PSD digest
    digest = out.to(torch.float64).mean(dim=(1, 2)).reshape(-1, 1)
    digest_all = distributed.gather_channel_summaries(digest, n_channels)

    octave = octave_band_leg(dev, world, rank) if n_fft == 1024 else None

    if rank == 0:
        spectra_per_step = n_channels * F
        value = spectra_per_step * args.steps / elapsed
        bytes_per_spectrum = 4 * hop + 4 * (n_fft // 2 + 1)
        bytes_per_launch = len(my_channels) * F * bytes_per_spectrum
        achieved = bytes_per_launch / (kernel_ms_max * 1e-3) / 1e9
        result = {
            "metric": "spectra/sec (1024-pt STFT)" if n_fft == 1024 else f"spectra/sec ({n_fft}-pt STFT)",
            "value": value,
            "unit": "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"rolling spectrogram: {n_fft}-pt Hann STFT, hop {hop}, "
                                   f"{'dB + A-weighting + colour LUT -> u32 pixels' if kind == 3 else args.kind}, "
                                   f"{cpg} ch/GPU x 2^{args.log2_samples} samples @ 48 kHz "
                                   f"(BASELINE configs[1])",
                       "channels": n_channels, "spectra_per_step": spectra_per_step, "batches_rotated": nbatch,
                       "parallelism": f"channel-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "stft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(n_fft, hop, F),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel_ms": kernel_ms_max},
            "digest": float(digest_all.sum().item()),
        }
        if psd_ms is not None:
            result["psd_output"] = {"kernel_ms": psd_ms, "spectra_per_s": spectra_per_step / (psd_ms * 1e-3),
                                    "frac_of_hbm_peak": bytes_per_launch / (psd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "same launches writing the float PSD instead of colour pixels (same byte counts)"}
        if same_batch_ms is not None:
            result["same_batch"] = {"kernel_ms": same_batch_ms, "spectra_per_s": spectra_per_step / (same_batch_ms * 1e-3) / 1.0,
                                    "note": "the same batch every step (what a naive loop measures); with the default plain row "
                                            "stores the rows evict the samples and this equals the rotating figure, with the "
                                            "-DFRT_NT_STORES build the samples partly survive in the Infinity Cache; not the headline"}
        if octave is not None:
            result["octave_bands"] = octave
        if world == 1 and args.cpu_budget > 0:
            result["cpu_baseline"] = cpu_baseline(host_x[0], n_fft, hop, consts["weight"], consts["lut"], args.cpu_budget)
        print(json.dumps(result), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
