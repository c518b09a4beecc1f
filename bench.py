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
steps (reported separately as `same_batch`).  Before anything is timed the GPU is brought to its
sustained clocks with --prewarm-ms of the same launches: straight after start-up a 55-launch run
sits inside the clock ramp and reads 17 % slow.  With N GPUs every rank owns its own channel(s)
(weak scaling, no data-path collective); `value` is the whole-job spectra/s: total spectra of all
ranks / max-over-ranks time.

One JSON line is printed by rank 0; besides the contract fields it carries
  roofline      HBM roofline of the dominant kernel (stft_kernel): algorithmic bytes per launch
                (4*hop + 4*(N/2+1) = 4100 B per spectrum) / average launch duration measured with
                HIP events on the launch stream, against the 8 TB/s peak
  parity        the colour image of the timed batch against the oracle on a sample of frames:
                pixels checked / differing, split into "the epilogue given its float32 PSD"
                (exact) and "float32 PSD moved a bin across an index edge"
  cpu_baseline  the oracle (numpy restatement of the reference path, 1 core and all cores) timed on
                a bounded sample of the same input on this box's host
  legs          the other BASELINE configurations on this GPU's shard, each with its own roofline:
                configs[2] 1/3-octave bank 8 ch (exact IIR time-parallel, exact IIR sequential
                = the bit-exact mode, FIR overlap-add bank), configs[3] 16384-point STFT 32 ch,
                configs[4] GCC-PHAT 100 window pairs + 1/24-octave bank 8 ch
  ranks_seen    the ranks an all-gather over the job's process group returned (RCCL on GPUs); the job exits non-zero
                unless they are exactly 0..N-1
  per_rank      every rank's own wall time per step (min / max / all): what the max-over-ranks hides
  slab_gather   with --gather-slabs: the optional all-gather of the output slabs, issued asynchronously
                behind each step and overlapped with the next one; `value` then includes its cost
Exit status: 0, or 3 when a parity gate of the timed batch fails (the JSON line is printed either way).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
F64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X float64 vector peak (the matrix rate is the same on this chip)
SIMDS, MAX_CLOCK_HZ = 1024, 2.4e9


def synth_channel(channel: int, n: int) -> np.ndarray:
    """S-noise of SURVEY.md §8d: 0.25 * standard_normal, seed 42 + channel, float32 PCM."""
    return (0.25 * np.random.default_rng(42 + channel).standard_normal(n, dtype=np.float32)).astype(np.float32)


def kernel_source_digest() -> str:
    """Digest of the sources of the headline kernel (stft_kernel, N <= 1024: stft_wave.h + fft_core.h, and stft_launch of stft.hip,
    which picks its instance, run length and grid; stft_big.h / stft_pk*.h hold the N >= 2048 instances) — their code, i.e. with comments and blank lines removed: a PMC
    traffic figure is only quoted for the code it was measured on (rewording a comment does not un-measure it).
    tests/test_evidence_fresh.py fails while profiles/pmc_traffic.json carries another digest."""
    import re
    h = hashlib.sha256()

    def code(text):
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        lines = [re.sub(r"//[^\n]*", "", ln).rstrip() for ln in text.splitlines()]
        return "\n".join(ln for ln in lines if ln.strip()).encode()
    for name in ("stft_wave.h", "fft_core.h"):
        h.update(code((ROOT / "friture_amd" / "csrc" / name).read_text()))
    # ... and the launch geometry (run length, lane groups, instance selection: it decides the re-read share of a launch)
    host = (ROOT / "friture_amd" / "csrc" / "stft.hip").read_text()
    m = re.search(r"static int stft_launch\(.*?\n}\n", host, re.S)
    h.update(code(m.group(0)) if m else b"stft_launch not found")
    return h.hexdigest()[:16]


# Kernel sources behind each leg: a leg's PMC traffic figure (profiles/r06_leg_traffic.json, tools/leg_traffic.py) is quoted only while
# the code of these files is what it was measured on.
LEG_SOURCES = {
    "configs4_gcc_phat_1024_pairs": ["gcc.hip", "gcc_resident.h", "fft_static.h", "fft_mixed.h"],
    "configs4_gcc_phat": ["gcc.hip", "gcc_resident.h", "fft_static.h", "fft_mixed.h"],
    "configs2_bank_iir_time_parallel": ["iir.hip", "octbank.h"],
    "configs4_bank_iir_time_parallel": ["iir.hip", "octbank.h"],
    "configs2_bank_fir_overlap_add": ["ola.hip", "ola_wave.h", "octbank.h"],
    "configs4_bank_fir_overlap_add": ["ola.hip", "ola_wave.h", "octbank.h"],
    "configs3_stft16384_psd": ["stft_pk16.h", "stft_pk.h", "fft_core.h"],
    "configs3_stft16384_image": ["stft_pk16.h", "stft_pk.h", "fft_core.h"],
    "configs3_stft16384_hop4096_psd": ["stft_pk16.h", "stft_pk.h", "fft_core.h"],
    "configs3_stft16384_hop4096_image": ["stft_pk16.h", "stft_pk.h", "fft_core.h"],
    "configs1_f64_psd": ["stft_wave.h", "fft_core.h"],
    "configs1_f64_image": ["stft_wave.h", "fft_core.h"],
}


def sources_digest(names) -> str:
    """sha256 of the code (comments and blank lines removed) of the named files under friture_amd/csrc, 16 hex digits."""
    import re
    h = hashlib.sha256()
    for name in names:
        text = (ROOT / "friture_amd" / "csrc" / name).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        lines = [re.sub(r"//[^\n]*", "", ln).rstrip() for ln in text.splitlines()]
        h.update("\n".join(ln for ln in lines if ln.strip()).encode())
    return h.hexdigest()[:16]


_LEG_TRAFFIC = None


def leg_traffic(leg_name: str):
    """HBM bytes per step of a leg's kernels from the committed PMC record (profiles/r06_leg_traffic.json), or None when the record
    is missing or was measured on other kernel sources (tests/test_evidence_fresh.py keeps it fresh)."""
    global _LEG_TRAFFIC
    if _LEG_TRAFFIC is None:
        try:
            _LEG_TRAFFIC = json.loads((ROOT / "profiles" / "r06_leg_traffic.json").read_text()).get("legs", {})
        except Exception:
            _LEG_TRAFFIC = {}
    for rec in _LEG_TRAFFIC.values():
        if leg_name in rec.get("bench_legs", []) and rec.get("kernel_sources") == sources_digest(LEG_SOURCES[leg_name]):
            return rec.get("hbm_bytes_per_call")
    return None


# Keys whose values are prose: they say how a figure was obtained, which DESIGN.md §6 and this file's docstrings also say.  They
# stay in the full record (--full-json) and leave the printed line, which the driver keeps only the last ~8 KB of.
PROSE_KEYS = frozenset({"mode", "model", "f64_note", "note", "traffic_note", "bytes_model"})
LINE_BUDGET = 7000


def compact_line(result: dict) -> dict:
    """The record as printed: no prose keys, floats at 5 significant digits, `octave_bands` as a pointer to its leg plus the
    contract-style figures (the leg itself is in `legs`).  Everything numeric survives; tests/test_bench_line.py holds the
    printed line of a full run (every leg) under LINE_BUDGET characters."""
    def walk(o):
        if isinstance(o, dict):
            return {k: walk(v) for k, v in o.items() if k not in PROSE_KEYS}
        if isinstance(o, (list, tuple)):
            return [walk(v) for v in o]
        if isinstance(o, float):
            return float(f"{o:.5g}")
        return o
    out = walk(result)
    for k in ("value", "ms_per_step"):              # the contract's own figures keep their digits (value x ms_per_step is checked)
        if k in result:
            out[k] = result[k]
    roof_keys = ("bound", "kernel", "achieved", "frac", "traffic", "frac_on_survey_bytes", "kernel_ms", "hbm_frac", "f64_frac")      # (unit and peak follow from `bound`)
    parity_keys = ("frames_checked", "pixels_mismatched", "mismatch_unaccounted", "epilogue_mismatch_outside_edge",
                   "psd_rel_max", "gate")

    def base(b):
        short = {k: b[k] for k in ("value", "cores") if k in b}          # (unit = the leg's, kind = the headline's; `sample`: --full-json)
        if isinstance(b.get("all_cores"), dict):
            short["all_cores"] = {k: b["all_cores"][k] for k in ("value", "cores") if k in b["all_cores"]}
        return short
    for lg in (out.get("legs") or {}).values():
        if isinstance(lg.get("roofline"), dict):
            lg["roofline"] = {k: lg["roofline"][k] for k in roof_keys if k in lg["roofline"]}
        if isinstance(lg.get("parity"), dict):
            lg["parity"] = {k: lg["parity"][k] for k in parity_keys if k in lg["parity"]}
        if isinstance(lg.get("cpu_baseline"), dict):
            lg["cpu_baseline"] = base(lg["cpu_baseline"])
    if isinstance(out.get("parity"), dict):
        out["parity"].pop("layout", None)             # (says what config.layout says)
    ob = out.get("octave_bands")
    if isinstance(ob, dict) and "legs" in out:
        out["octave_bands"] = {"leg": "configs2_bank_iir_time_parallel", "value": ob.get("value"), "unit": ob.get("unit"),
                               "ms_per_step": ob.get("ms_per_step")}
    return out


def cpu_baseline(x: np.ndarray, n_fft: int, hop: int, weight, lut, budget_s: float, with_legs: bool = True):
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
              # kind = port: the reference checkout is not on the bench box; oracle/dsp.py is its numpy float64 restatement
              # (analyzelive + dB + A-weighting + colour LUT), bit-exact against it (tests/golden) and the faster of the two
              "sample": f"{passes} pass(es) x first {frames} of {frames_total} spectra of ch 0 (same input), {dt:.1f} s, oracle/dsp.py",
              "host_cpus": os.cpu_count()}
    # every core: one spawned worker per core (workers import numpy + the oracle only, never the GPU runtime); the same pool
    # then times the other halves of the metric — the octave banks (both variants, bpo 3 and 24) and GCC-PHAT — whose
    # one-core figures are taken here in the parent first
    side = {}
    sc = min(1.0, budget_s / 12.0)                         # about 4.5 s of one-core work at the default budget
    nb3, nb24, nwin = max(16, int(1200 * sc)), max(16, int(256 * sc)), max(8, int(256 * sc))
    side_jobs = {"octave_iir_bpo3": (cpu_bench.octave_blocks, (0, 3, nb3, "iir"), "octave-bands/s"),
                 "octave_ola_bpo3": (cpu_bench.octave_blocks, (0, 3, nb3, "ola"), "octave-bands/s"),
                 "octave_iir_bpo24": (cpu_bench.octave_blocks, (0, 24, nb24, "iir"), "octave-bands/s"),
                 "octave_ola_bpo24": (cpu_bench.octave_blocks, (0, 24, nb24, "ola"), "octave-bands/s"),
                 "gcc_phat": (cpu_bench.gcc_windows, (4242, 24000, nwin), "windows/s")} if with_legs else {}
    for name, (fn, job, unit) in side_jobs.items():
        t0 = time.perf_counter()
        units, _ = fn(job)
        dt1 = time.perf_counter() - t0
        # octave legs: the exact IIR bank through oracle/iir_ref.c or the FFT overlap-add bank in numpy, + smoothed band energies
        # + dB per block; GCC-PHAT: numpy rfft / irfft float64 (friture/signal/correlation.py:24-43) + arg-max
        sample = (f"{job[2]} blocks x 1024 samples, 1 ch, bpo {job[1]}, {job[3]}, {dt1:.2f} s" if fn is cpu_bench.octave_blocks else
                  f"{job[2]} window pairs, L = {job[1]}, {dt1:.2f} s")
        side[name] = {"value": units / dt1, "unit": unit, "cores": 1, "kind": "port", "sample": sample}
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
                                   "sample": f"{cores} processes x {wframes} spectra, {wall:.1f} s"}
            for name, (fn, sjob, unit) in side_jobs.items():
                sjob = (sjob[0], sjob[1], max(8, sjob[2] // 4), *sjob[3:])
                t0 = time.perf_counter()
                done = list(pool.map(fn, [sjob] * cores))
                wall = time.perf_counter() - t0
                side[name]["all_cores"] = {"value": sum(d[0] for d in done) / wall, "unit": unit, "cores": cores,
                                           "sample": f"{cores} processes x {sjob[2]} units, {wall:.2f} s"}
    except Exception as exc:                                                     # a reported extra, never fatal
        result.setdefault("all_cores", {"error": repr(exc)})
    result["_side"] = side
    return result


def image_parity_report(eng, x, image, n_fft, hop, weight, lut, frames=4096):
    """The timed batch's colour image against the oracle on its first `frames` frames of channel 0 (outside the timed
    region).  Three ingredients: the image the timed launches wrote, the float32 PSD the same kernel produces for the
    same samples (PSD kind), and the oracle's float64 PSD + epilogue."""
    import torch
    from oracle import dsp
    frames = min(frames, image.shape[1])
    T = n_fft + hop * (frames - 1)
    xs = x[:1, :T].contiguous()
    from friture_amd.stft import StftEngine
    e1 = StftEngine(n_fft, hop, 1, 32)
    e1.set_epilogue(weight, -140.0, 0.0, lut)
    psd32 = e1.psd(xs)[0].cpu().numpy()
    img = image[0, :frames].cpu().numpy().view(np.uint32)
    torch.cuda.synchronize()
    psd64 = dsp.stft_psd(xs[0].cpu().numpy().astype(np.float64), n_fft, hop)
    rep = dsp.image_parity(img, psd32, psd64, weight, -140.0, 0.0, lut)
    rep["frames_checked"] = frames
    # the gates: the epilogue exact given its own PSD; every pixel that differs from the float64 reference image explained
    # by the measured float32 PSD error at that bin (oracle/dsp.py:image_parity); the PSD within the 1e-5 bar
    rep["gate"] = {"epilogue_exact": rep["epilogue_mismatch_outside_edge"] == 0,
                   "every_mismatch_accounted": rep["mismatch_unaccounted"] == 0,
                   "psd_within_1e-5": rep["psd_rel_max"] <= 1e-5}
    rep["gate"]["pass"] = all(rep["gate"].values())
    rep["note"] = ("epilogue_*: GPU pixels against the reference's float64 dB -> normalise -> index -> LUT applied to the GPU's "
                   "own float32 PSD (exact outside 1e-6 of an index edge); pixels_mismatched: against the float64 reference "
                   "image, i.e. bins the float32 PSD error (psd_rel_max of the frame maximum) carried across an edge")
    return rep


def timed(fn, steps, dev, distributed, torch):
    """`steps` calls of fn bracketed by barrier + synchronize; returns (wall seconds, HIP-event ms per call)."""
    solo = distributed.world_is_one()             # a job of one process has nobody to wait for: no barrier call, one synchronize a side
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync = torch.cuda.synchronize
    torch.cuda.synchronize()
    if not solo:
        distributed.barrier(dev)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(steps):
        fn(k)
    ev1.record()
    sync()
    if not solo:
        distributed.barrier(dev)
        sync()
    t1 = time.perf_counter()
    return t1 - t0, ev0.elapsed_time(ev1) / steps


def prewarm(fn, seconds, torch, batch=4):
    """Untimed launches for `seconds` before the warm-up steps (the clock ramp), `batch` of them between two synchronisations.
    Measured (profiles/r06_prewarm_ab.txt, headline only, alternating runs on one box): batches of 4 (the device idles for the
    host's wake-up every 0.45 ms) give value 1.153 / 1.126 / 1.148 x10^9, batches of 32 (the queue never runs dry) 1.133 / 1.130 /
    1.133 — the hotter pre-warm costs the timed region about a percent of clock; 4 stays."""
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(batch):
            fn(k)
            k += 1
        torch.cuda.synchronize()


def leg(fn, steps, dev, distributed, torch, prewarm_s=0.15):
    prewarm(fn, prewarm_s, torch)
    wall, ev_ms = timed(fn, steps, dev, distributed, torch)
    return distributed.max_over_ranks(wall, dev) / steps, distributed.max_over_ranks(ev_ms, dev)


def octave_legs(dev, world, rank, ch, bpo, log2n, tag, with_sequential):
    """octave-bands/s of the three bank variants on `ch` channels per GPU x 2^log2n samples, energies per 1024 samples."""
    import torch

    from friture_amd import distributed, filter_design
    from friture_amd.filter import FirBank, IirBank
    t = filter_design.load_tables()
    n = 1 << log2n
    x = torch.from_numpy(np.stack([synth_channel(1000 + rank * ch + c, n) for c in range(ch)])).to(dev)
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])     # octavespectrum.py:145-153
    out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
    units = world * ch * (n // 1024) * 9 * bpo
    alg_bytes = ch * (n // 1024) * (4096 + 4 * 9 * bpo)
    # instruction floor of the output pass (round 3: one lane per (channel, chunk), the whole state vector in registers, wave-
    # uniform coefficients as scalar operands): 2 * 4 + 1 + 2 = 11 float64 instructions per sample and band filter (9 of the
    # recurrence, 2 of the block energy), 2 * 12 + 1 = 25 per sample for the decimator, 64 chunks per wavefront; a float64
    # vector instruction occupies its SIMD for 4 cycles
    wave_instr = sum((n >> j) * ch * (11 * bpo + 25) / 64 for j in range(9))
    issue_bound_s = wave_instr * 4 / (SIMDS * MAX_CLOCK_HZ)
    legs = {}
    # SURVEY.md §8d's algorithmic flops of the bank (mode 0): (2 bpo + 6) biquads x 2044 samples x 9 flop per channel and
    # 1024-sample block (220.8 kflop at bpo 3, 993 kflop at bpo 24) — the figure every variant is priced against
    survey_flops = ch * (n // 1024) * (2 * bpo + 6) * 2044 * 9

    def record(name, bank, steps, mode, extra):
        dt, ev_ms = leg(lambda k: bank.energies(x, 1024, alphas, out=out), steps, dev, distributed, torch)
        legs[f"{tag}_{name}"] = {"value": units / dt, "unit": "octave-bands/s", "ms_per_step": dt * 1e3, "mode": mode,
                                 "config": f"{ch} ch x 2^{log2n}, {9 * bpo} bands, energies / 1024 samples",
                                 "roofline": dict({"algorithmic_bytes_per_step": alg_bytes, "traffic": leg_traffic(f"{tag}_{name}"),
                                                   "hbm_frac": alg_bytes / dt / 1e9 / HBM_PEAK_GBS,
                                                   "algorithmic_flops_per_step": survey_flops,
                                                   "f64_tflops": survey_flops / dt / 1e12,
                                                   "f64_frac": survey_flops / dt / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                                                   "f64_note": "SURVEY.md §8d flop model of the exact bank ((2 bpo + 6) biquads x 2044 "
                                                               "samples x 9 flop per channel-block) over this variant's time, against "
                                                               "the 78.6 TFLOP/s float64 vector peak"}, **extra(dt))}

    iir = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
    chunk = 1024 if bpo <= 3 else 512      # measured 2048 / 1024 / 512 / 256 (tools/exp/chunk_sweep.sh): 0.74 / 0.62 / 0.64 / 0.76 ms at bpo 3, 0.82 / 0.66 / 0.64 / 0.75 at bpo 24
    iir.set_chunk(chunk)
    record("iir_time_parallel", iir, 10,
           f"exact IIR bank, time-parallel chunks of {chunk} samples: the same recurrences re-associated; output pass with one lane "
           f"per (channel, chunk) and contracted multiply-adds (energy-only call); band energies equal to the bit-exact "
           f"sequential mode's at float32 output precision (bar 1e-5)",
           lambda dt: {"bound": "f64_valu_issue", "unit": "s", "achieved": dt, "peak": issue_bound_s, "frac": issue_bound_s / dt,
                       "model": "sum over stages of samples x channels x (11 bpo + 25) float64 instructions / 64 lanes x 4 cycles / "
                                "(1024 SIMDs x 2.4 GHz): the arithmetic of the output pass alone; the time-parallel mode adds the "
                                "zero-state products (matrix cores) and the chunk scans on top"})
    if with_sequential:
        seq = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
        xs, outs = x[:, : 1 << 16].contiguous(), torch.empty((ch, 64, 9 * bpo), dtype=torch.float32, device=dev)
        dt, _ = leg(lambda k: seq.energies(xs, 1024, alphas, out=outs), 3, dev, distributed, torch)
        legs[f"{tag}_iir_sequential"] = {"value": world * ch * 64 * 9 * bpo / dt, "unit": "octave-bands/s", "ms_per_step": dt * 1e3,
                                         "mode": "exact IIR bank, sequential in time: bit-identical to the reference's recurrence",
                                         "config": f"{ch} ch x 2^16"}
    fir = FirBank(bpo, ch, t)
    nfilt = bpo + 1
    tiles = sum(-(-(n >> j) // 3072) for j in range(9)) * ch
    flops = tiles * (1 + nfilt) * (5 * 2048 * 11 + 16 * 2048)       # complex FFTs of 2048 points + pack / unpack / multiply
    record("fir_overlap_add", fir, 10,
           "FFT overlap-add bank (the reference's production bank, 512-tap FIRs): batched, 1e-11 of the band maximum from the "
           "reference fed in 1024-sample blocks",
           lambda dt: {"bound": "f64_flops", "unit": "TFLOP/s", "achieved": flops / dt / 1e12, "peak": F64_VECTOR_PEAK_TFLOPS,
                       "frac": flops / dt / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                       "model": "tiles of 3072 outputs x (1 forward + bpo + 1 inverse) complex FFTs of 2048 points x "
                                "(5 M log2 M + 16 M) float64 operations"})
    return legs


def stft16384_leg(dev, world, rank, consts):
    """configs[3]: 256-ch batched spectrogram with 16384-point frames, 32 channels per GPU x 2^20 samples, at hop N/2 (the
    overlap BASELINE's byte figure is quoted on) and hop N/4 (75 % overlap, the default of both reference widgets:
    friture/spectrogram.py:91-93, friture/spectrum.py:66)."""
    import torch

    from friture_amd import distributed, tables
    from friture_amd.stft import StftEngine
    n_fft, ch, T = 16384, 32, 1 << 20
    weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
    xs = [torch.from_numpy(np.stack([synth_channel(5000 + 977 * b + rank * ch + c, T) for c in range(ch)])).to(dev) for b in range(3)]
    legs = {}
    for hop, tag in ((8192, ""), (4096, "_hop4096")):
        eng = StftEngine(n_fft, hop, ch, 32)
        eng.set_epilogue(weight, -140.0, 0.0, consts["lut"])
        F = eng.frames_for(T)
        bytes_per_launch = ch * F * (4 * hop + 4 * (n_fft // 2 + 1))
        for kind, name in ((3, "image"), (0, "psd")):
            outs = [torch.empty((ch, F, n_fft // 2 + 1), dtype=torch.int32 if kind == 3 else torch.float32, device=dev) for _ in range(3)]
            dt, ev_ms = leg(lambda k: eng.run(kind, xs[k % 3], outs[k % 3]), 30, dev, distributed, torch)
            legs[f"configs3_stft16384{tag}_{name}"] = {
                "value": world * ch * F / dt, "unit": "spectra/s", "ms_per_step": dt * 1e3,
                "config": f"{ch} ch x 2^20, N {n_fft}, hop {hop}, {'pixels' if kind == 3 else 'PSD'}",
                "roofline": {"bound": "hbm", "kernel": "stft_pk16_kernel", "unit": "GB/s", "achieved": bytes_per_launch / (ev_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "frac": bytes_per_launch / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": leg_traffic(f"configs3_stft16384{tag}_{name}"),
                             "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": ev_ms}}
            del outs
        del eng
    return legs


def stft1024_f64_leg(dev, world, rank, consts):
    """configs[1] at the reference's own precision: float64 samples in (audio promoted at friture/audiobackend.py:466-468),
    float64 PSD out (friture/audioproc.py:42-50) — 8 * 512 + 8 * 513 = 8200 bytes per spectrum — and the float64 colour kind
    (u32 pixels out, 6148 bytes per spectrum), whose image must be the float64 reference's image pixel for pixel."""
    import torch

    from friture_amd import distributed
    from friture_amd.stft import StftEngine
    from oracle import dsp
    n_fft, hop, T = 1024, 512, 1 << 25
    nb = n_fft // 2 + 1
    xs = [torch.from_numpy(np.stack([synth_channel(7000 + 131 * b + rank, T).astype(np.float64)])).to(dev) for b in range(3)]
    eng = StftEngine(n_fft, hop, 1, 64)
    eng.set_epilogue(consts["weight"], -140.0, 0.0, consts["lut"])
    F = eng.frames_for(T)
    legs = {}
    for kind, name, out_bytes in ((0, "psd", 8), (3, "image", 4)):
        # split rows, as the headline: [F][N/2] + Nyquist plane [F] in one allocation per batch
        odt = torch.int32 if kind == 3 else torch.float64
        slabs = [torch.empty((F * nb,), dtype=odt, device=dev) for _ in range(3)]
        rows = [sl[:F * (nb - 1)].view(1, F, nb - 1) for sl in slabs]
        nyqs = [sl[F * (nb - 1):].view(1, F) for sl in slabs]
        dt, ev_ms = leg(lambda k: eng.run_split(kind, xs[k % 3], rows[k % 3], nyqs[k % 3]), 30, dev, distributed, torch)
        outs = [torch.cat([rows[0], nyqs[0][..., None]], dim=2)]              # the packed shape the checks below are written for
        bytes_per_launch = F * (8 * hop + out_bytes * nb)
        rec = {"value": world * F / dt, "unit": "spectra/s", "ms_per_step": dt * 1e3, "dtype": "f64",
               "config": f"1 ch x 2^25, N {n_fft}, hop {hop}, f64 -> {'pixels' if kind == 3 else 'f64 PSD'}, split rows",
               "roofline": {"bound": "hbm", "kernel": "stft_kernel<double>", "unit": "GB/s", "achieved": bytes_per_launch / (ev_ms * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "frac": bytes_per_launch / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": leg_traffic(f"configs1_f64_{name}"),
                            "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_per_spectrum": 8 * hop + out_bytes * nb,
                            "kernel_ms": ev_ms}}
        if rank == 0:
            frames = 2048
            Ts = n_fft + hop * (frames - 1)
            ref = dsp.stft_psd(xs[0][0, :Ts].cpu().numpy(), n_fft, hop)
            got = outs[0][0, :frames].cpu().numpy()
            if kind == 0:
                err = float(np.max(np.max(np.abs(got - ref), axis=1) / np.max(ref, axis=1)))
                rec["parity"] = {"frames_checked": frames, "psd_rel_max": err, "gate": {"psd_within_1e-12": err <= 1e-12, "pass": err <= 1e-12}}
            else:
                psd_gpu = eng.psd(xs[0][:, :Ts].contiguous())[0].cpu().numpy()
                rep = dsp.image_parity(got.view(np.uint32), psd_gpu, ref, consts["weight"], -140.0, 0.0, consts["lut"], edge=1e-9)
                rep["frames_checked"] = frames
                rep["gate"] = {"epilogue_exact": rep["epilogue_mismatch_outside_edge"] == 0,
                               "pixel_exact_outside_1e-9_of_an_edge": rep["mismatch_outside_edge"] == 0}
                rep["gate"]["pass"] = all(rep["gate"].values())
                rec["parity"] = rep
        legs[f"configs1_f64_{name}"] = rec
        del outs, slabs, rows, nyqs
    return legs


def gcc_leg(dev, world, rank):
    """configs[4], first half: GCC-PHAT of 100 window pairs of L = 24000 samples (float64) per GPU — and of 1024 pairs, the
    batch at which every CU holds a pair."""
    import torch

    from friture_amd import distributed
    from friture_amd.signal.correlation import GccPhat
    L = 24000
    out = {}
    for pairs, name in ((100, "configs4_gcc_phat"), (1024, "configs4_gcc_phat_1024_pairs")):
        rng = np.random.default_rng(4242 + rank)
        d0 = 0.25 * rng.standard_normal((pairs, L))
        d1 = np.roll(d0, 37, axis=1) + 0.025 * rng.standard_normal((pairs, L))
        g = GccPhat(L, pairs)
        a0, a1 = torch.from_numpy(d0).to(dev), torch.from_numpy(d1).to(dev)
        dt, ev_ms = leg(lambda k: g.correlate(a0, a1), 20, dev, distributed, torch)
        _, am = g.correlate(a0, a1)
        nbytes = pairs * 24 * L
        # real transforms of 24000 samples as complex transforms of 12000: 3 per pair (two forward, one inverse), 5 n log2 n
        flops = pairs * 3 * 5.0 * 12000 * np.log2(12000.0)
        out[name] = {"value": world * pairs / dt, "unit": "windows/s", "ms_per_step": dt * 1e3,
                     "config": f"{pairs} pairs, L {L}, f64", "delay_37_found": bool(int(am[0]) == 37),
                     "roofline": {"bound": "hbm", "kernel": "gcc_phat_resident_kernel", "unit": "GB/s", "achieved": nbytes / dt / 1e9,
                                  "peak": HBM_PEAK_GBS, "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS,
                                  "traffic": leg_traffic(name),
                                  "bytes_model": "24 L per window pair: two float64 windows read, one float64 correlation written (576 000 B); "
                                                 "SURVEY.md §8d's 288 000 B counts float32 samples, the reference computes in float64",
                                  "frac_on_survey_bytes": pairs * 12 * L / dt / 1e9 / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_step": nbytes, "algorithmic_flops_per_step": flops,
                                  "f64_tflops": flops / dt / 1e12, "f64_frac": flops / dt / 1e12 / F64_VECTOR_PEAK_TFLOPS}}
        del a0, a1, g
    return out


class StubEngine:
    """CPU stand-in for StftEngine, selected by --stub-engine: lets the rank logic of this file (sharding, table
    broadcast, barriers, max-over-ranks timing, digest gather) run end to end over gloo in tests/.  It computes nothing
    of the hot path (every 'pixel' is the channel's first sample scaled) and its JSON line says data = "stub"."""

    def __init__(self, n_fft, hop, n_channels, precision):
        self.n_fft, self.hop, self.n_channels = n_fft, hop, n_channels

    def set_epilogue(self, *a):
        pass

    def frames_for(self, T):
        return (T - self.n_fft) // self.hop + 1

    def run(self, kind, x, out):
        out[:] = (x[:, :1, None] * 1000).to(out.dtype)
        return out

    def run_split(self, kind, x, rows, nyq):
        rows[:] = (x[:, :1, None] * 1000).to(rows.dtype)
        nyq[:] = (x[:, :1] * 1000).to(nyq.dtype)
        return rows, nyq


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` started without a launcher (WORLD_SIZE unset): re-execute the same command line as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>`
    — one rank per GPU, rank 0 prints the one JSON line — so both forms of the contract's launch work.  The port is a free
    one picked here; the child's exit status is this process's."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                     # (torchrun would set it, with a warning)
    raise SystemExit(subprocess.call(cmd, env=env))


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
    ap.add_argument("--layout", choices=["split", "packed"], default="split",
                    help="output rows: split = [F][N/2] rows on 64-byte boundaries + a Nyquist plane [F] (frt_stft_run_split, the batch "
                         "layout); packed = [F][N/2+1] (frt_stft_run, what the drop-in classes hand on).  Same values, same byte counts")
    ap.add_argument("--batches", type=int, default=3, help="distinct input batches the steps rotate over (1 = same batch every step)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0, help="untimed launches before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--prewarm-batch", type=int, default=4, help="launches queued between two synchronisations of the pre-warm")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for cpu_baseline (0 = skip)")
    ap.add_argument("--no-legs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--gather-slabs", action="store_true", help="all-gather every step's output slab over the job's process group, "
                    "asynchronously behind the step and overlapped with the next one (SURVEY.md §8e; optional: nobody needs "
                    "every channel's slab on every GPU by default)")
    ap.add_argument("--full-json", default="", help="also write the uncompacted record (prose notes, full-precision floats) to this file")
    ap.add_argument("--stub-engine", action="store_true", help="tests only: CPU stand-in engine over gloo, see StubEngine")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)        # plain `python bench.py --gpus N`: becomes the contract's launch line (does not return)

    import torch

    from friture_amd import distributed

    stub = args.stub_engine
    rank, local_rank, world = distributed.init_process_group(backend="gloo" if stub else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if stub:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None                           # the stub has no device queue

        class _Ev:
            def __init__(self, **k):
                self.t = 0.0

            def record(self):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3
        torch.cuda.Event = _Ev
        Engine = StubEngine
        consts = {"weight": np.zeros(args.fft_size // 2 + 1), "lut": np.zeros(256, np.uint32)}
        if rank == 0:
            consts = {"weight": np.linspace(-30.0, 2.0, args.fft_size // 2 + 1), "lut": np.arange(256, dtype=np.uint32) | 0xFF000000}
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
        from friture_amd import _lib, palette, tables
        from friture_amd.stft import StftEngine as Engine
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        _lib.init(local_rank)
        consts = {"weight": np.zeros(args.fft_size // 2 + 1), "lut": np.zeros(256, np.uint32)}
        if rank == 0:
            consts["weight"] = tables.weighting_db(tables.rfft_frequencies(args.fft_size), 1e-50)[0]
            consts["lut"] = palette.cmr_lut()

    n_fft = args.fft_size
    hop = args.hop or n_fft // 2
    T = 1 << args.log2_samples
    cpg = args.channels_per_gpu
    n_channels = cpg * world
    my_channels = distributed.shard_channels(n_channels, rank, world)
    # constant tables: computed on rank 0, broadcast (RCCL) so every rank uses identical bits
    consts = distributed.broadcast_tables(consts, src=0, device=dev)
    ranks_seen = distributed.gather_ranks(dev)

    host_x = [synth_channel(c, T) for c in my_channels]
    nbatch = max(1, args.batches)
    xs = [torch.from_numpy(np.stack(host_x)).to(dev)]
    for b in range(1, nbatch):                                  # further batches: same statistics, other seeds
        xs.append(torch.from_numpy(np.stack([synth_channel(100000 * b + c, T) for c in my_channels])).to(dev))
    eng = Engine(n_fft, hop, len(my_channels), 32)
    eng.set_epilogue(consts["weight"], -140.0, 0.0, consts["lut"])
    kind = {"image": 3, "psd": 0, "db": 1}[args.kind]
    F = eng.frames_for(T)
    split = args.layout == "split" and n_fft <= 1024 and not args.gather_slabs      # (the slab gather's check is written for packed slabs)
    odt = torch.int32 if kind == 3 else torch.float32
    if split:
        # one allocation per batch: the rows (whole 64-byte lines each), then the Nyquist plane
        slabs = [torch.empty((len(my_channels) * F * (n_fft // 2 + 1),), dtype=odt, device=dev) for _ in range(nbatch)]
        nrow = len(my_channels) * F * (n_fft // 2)
        rows = [sl[:nrow].view(len(my_channels), F, n_fft // 2) for sl in slabs]
        nyqs = [sl[nrow:].view(len(my_channels), F) for sl in slabs]
        outs = slabs
    else:
        outs = [torch.empty((len(my_channels), F, n_fft // 2 + 1), dtype=odt, device=dev) for _ in range(nbatch)]

    gather = None
    if args.gather_slabs:
        if nbatch < 2:
            raise SystemExit("--gather-slabs needs --batches >= 2 (a slab is gathered while the next batch is computed)")
        gather = distributed.SlabGather(outs[0], n_slots=nbatch)

    vdt = [odt]                                   # element type the buffers are viewed as (the PSD pass re-uses the image buffers)

    # The launches of the steps, prepared once per (output kind, batch): StftEngine.prepare checks shapes / types and extracts the
    # pointers here, a step is then ONE frt_stft_run(_split) call (the per-step .view() / pointer extraction
    # hoisted out of step()).  The stub engine has no prepare(): its steps call run() / run_split() as before.
    prepared = {}

    def launch_for(b):
        key = (kind, vdt[0], b)
        if key not in prepared:
            if stub:
                if split:
                    prepared[key] = lambda: eng.run_split(kind, xs[b], rows[b].view(vdt[0]), nyqs[b].view(vdt[0]))
                else:
                    prepared[key] = lambda: eng.run(kind, xs[b], outs[b].view(vdt[0]))
            elif split:
                prepared[key] = eng.prepare(kind, xs[b], rows[b].view(vdt[0]), nyqs[b].view(vdt[0]))
            else:
                prepared[key] = eng.prepare(kind, xs[b], outs[b].view(vdt[0]))
        return prepared[key]

    def step(k, rotate=True):
        b = k % nbatch if rotate else 0
        if gather is not None:
            gather.wait(b)                        # the previous gather of this buffer has read it
        launch_for(b)()                           # one kernel launch on torch's current stream
        if gather is not None:
            gather.start(outs[b], b)              # behind the launch above, beside the next step's

    # Clock ramp: after idle the GPU needs tens of milliseconds of continuous work before it runs at its sustained
    # clocks; 55 launches (8 ms) straight after start-up read ~17 % slow.  Pre-warm with the same launches for
    # --prewarm-ms (default 300 ms, untimed), then the contract's W warm-up steps, then the K timed steps.
    if not stub:
        prewarm(step, args.prewarm_ms * 1e-3, torch, args.prewarm_batch)
    for k in range(args.warmup):
        step(k)
    # One HIP event pair brackets the K launches on the launch stream (torch's current stream, which the C ABI launches
    # on): average launch duration = elapsed / K.  Repeated three times for the run-to-run spread; `value` is the FIRST.
    wall, kernel_ms = timed(step, args.steps, dev, distributed, torch)
    if gather is not None:
        gather.wait_all()
    elapsed = distributed.max_over_ranks(wall, dev)
    per_rank_wall = distributed.gather_scalars(wall, dev)
    kernel_ms_max = distributed.max_over_ranks(kernel_ms, dev)
    repeats = [kernel_ms_max] + [distributed.max_over_ranks(timed(step, args.steps, dev, distributed, torch)[1], dev) for _ in range(2)]
    slab_check = None
    if gather is not None:
        # every rank's slab arrived: the receive buffer of batch 0 against what each rank holds (first pixel / PSD value of
        # every channel, compared through the summary gather)
        got = gather.wait(0)
        firsts = distributed.gather_channel_summaries(outs[0][:, :1, 0].to(torch.float64), n_channels)
        slab_check = bool(torch.equal(got.reshape(-1, *outs[0].shape[1:])[:, 0, 0].to(torch.float64), firsts[:, 0]))
        gathered_bytes = gather.bytes_per_gather
        gather = None                              # the remaining measurements run without it
    same_batch_ms = None
    if nbatch > 1:
        same_batch_ms = distributed.max_over_ranks(timed(lambda k: step(k, False), args.steps, dev, distributed, torch)[1], dev)
    # the same launches into packed rows [F][N/2+1] (frt_stft_run, the layout the drop-in classes hand on): both layouts in one line
    packed_ms = None
    if split and not stub:
        packed = [torch.empty((len(my_channels), F, n_fft // 2 + 1), dtype=odt, device=dev) for _ in range(nbatch)]
        pstep = lambda k: eng.run(kind, xs[k % nbatch], packed[k % nbatch])       # noqa: E731
        for k in range(args.warmup):
            pstep(k)
        packed_ms = distributed.max_over_ranks(timed(pstep, args.steps, dev, distributed, torch)[1], dev)
        del packed
    # the same transform with its plain PSD output (no dB / weighting / colour epilogue), for reference
    psd_ms = None
    if kind == 3 and not stub:
        kind, vdt[0] = 0, torch.float32
        for k in range(args.warmup):
            step(k)
        psd_ms = distributed.max_over_ranks(timed(step, args.steps, dev, distributed, torch)[1], dev)
        kind, vdt[0] = 3, odt
        for b in range(nbatch):
            step(b)                               # the images back in place (digest, parity)
    if split:                                     # the timed batch's image in the packed shape the checks below are written for
        out = torch.cat([rows[0], nyqs[0][..., None]], dim=2)
    else:
        out = outs[0]

    # post-batch summary gather (outside the timed region): per-channel mean pixel/PSD digest
    digest = out.to(torch.float64).mean(dim=(1, 2)).reshape(-1, 1)
    digest_all = distributed.gather_channel_summaries(digest, n_channels)

    parity = None
    if rank == 0 and kind == 3 and not stub:
        parity = image_parity_report(eng, xs[0], out, n_fft, hop, consts["weight"], consts["lut"])
        parity["layout"] = "split rows + Nyquist plane of the timed launches, reassembled" if split else "packed"


    exit_code = 0
    legs = {}
    if not stub and not args.no_legs and n_fft == 1024:
        prepared.clear()                          # (the prepared launches hold their buffers)
        del xs, outs, out
        if split:
            del slabs, rows, nyqs
        torch.cuda.empty_cache()
        legs.update(stft1024_f64_leg(dev, world, rank, consts))
        legs.update(octave_legs(dev, world, rank, 8, 3, 22, "configs2_bank", True))
        legs.update(stft16384_leg(dev, world, rank, consts))
        legs.update(gcc_leg(dev, world, rank))
        legs.update(octave_legs(dev, world, rank, 8, 24, 20, "configs4_bank", False))

    # Every collective of the job is behind us: the process group goes away HERE, before rank 0 times the CPU baseline — the other
    # ranks are done and exit instead of spinning in a collective while rank 0 runs an all-cores pool on the same host.
    import torch.distributed as dist
    if dist.is_initialized():                  # (a group of one exists under FRT_DIST_FORCE=1: RCCL exercised on a single GPU)
        dist.destroy_process_group()

    if rank == 0:
        spectra_per_step = n_channels * F
        value = spectra_per_step * args.steps / elapsed
        bytes_per_spectrum = 4 * hop + 4 * (n_fft // 2 + 1)
        bytes_per_launch = len(my_channels) * F * bytes_per_spectrum
        achieved = bytes_per_launch / (kernel_ms_max * 1e-3) / 1e9
        traffic = None
        if not stub:
            p = ROOT / "profiles" / "pmc_traffic.json"
            try:
                rec = json.loads(p.read_text())
                if (rec.get("n_fft"), rec.get("hop"), rec.get("frames")) == (n_fft, hop, F) and rec.get("kernel_sources") == kernel_source_digest():
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                pass
        result = {
            "metric": "spectra/sec (1024-pt STFT)" if n_fft == 1024 else f"spectra/sec ({n_fft}-pt STFT)",
            "value": value,
            "unit": "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "layout": "split" if split else "packed",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "stub" if stub else "synthetic",
            "config": {"workload": f"rolling spectrogram: {n_fft}-pt Hann STFT, hop {hop}, "
                                   f"{'dB + A-weighting + colour LUT -> u32 pixels' if kind == 3 else args.kind}, "
                                   f"{cpg} ch/GPU x 2^{args.log2_samples} samples @ 48 kHz "
                                   f"(BASELINE configs[1]); rows "
                                   f"{'split [F][N/2] (64-byte lines) + Nyquist plane [F]: frt_stft_run_split' if split else 'packed [F][N/2+1]: frt_stft_run'}",
                       "layout": "split" if split else "packed",
                       "channels": n_channels, "spectra_per_step": spectra_per_step, "batches_rotated": nbatch,
                       "parallelism": f"channel-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "stft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_note": "HBM bytes per launch from the rocprofv3 PMC passes of this command (tools/gpu_session.sh -> "
                                         "profiles/pmc_traffic.json), quoted only when that file was measured on these kernel sources",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel_ms": kernel_ms_max, "kernel_ms_repeats": repeats,
                         # what the wall clock of the K timed steps holds beside the K launches' own time: the first launch's way to
                         # the device and the wake-up behind the last one (profiles/r06_host_fixed.txt: ~20 us per timed region,
                         # spinning on the completion signal instead of sleeping changes nothing); `value` stays on the wall clock
                         "host_fixed_us": (elapsed - kernel_ms_max * 1e-3 * args.steps) * 1e6},
            "ranks_seen": ranks_seen,
            "per_rank": {"ms_per_step_min": min(per_rank_wall) / args.steps * 1e3, "ms_per_step_max": max(per_rank_wall) / args.steps * 1e3,
                         "ms_per_step": [w / args.steps * 1e3 for w in per_rank_wall],
                         "note": "every rank's own wall time of the K timed steps between the two barriers; value uses the maximum"},
            "digest": float(digest_all.sum().item()),
        }
        if slab_check is not None:
            result["slab_gather"] = {"enabled": True, "bytes_per_step_per_rank": gathered_bytes, "slabs_verified": slab_check,
                                     "note": "all_gather_into_tensor of the step's output slab, async behind the step, overlapped with "
                                             "the next step (one receive buffer per rotating batch); included in value / ms_per_step"}
        if parity is not None:
            result["parity"] = parity
        if psd_ms is not None:
            result["psd_output"] = {"kernel_ms": psd_ms, "spectra_per_s": spectra_per_step / (psd_ms * 1e-3),
                                    "frac_of_hbm_peak": bytes_per_launch / (psd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "same batches, float PSD written instead of colour pixels (same byte counts); on aligned rows this kind runs the LDS-ring instance of the kernel, the colour kind the register-window instance — bit-identical spectra"}
        if packed_ms is not None:
            result["packed_rows"] = {"kernel_ms": packed_ms, "spectra_per_s": spectra_per_step / (packed_ms * 1e-3),
                                     "frac_of_hbm_peak": bytes_per_launch / (packed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "note": "same batches and kind through frt_stft_run: rows [F][N/2+1], the layout of rounds 1-4's headline and of the drop-in classes"}
        if same_batch_ms is not None:
            result["same_batch"] = {"kernel_ms": same_batch_ms, "spectra_per_s": spectra_per_step / (same_batch_ms * 1e-3) / 1.0,
                                    "note": "the same batch every step (what a naive loop measures); not the headline"}
        if legs:
            result["legs"] = legs
            result["octave_bands"] = legs.get("configs2_bank_iir_time_parallel")      # the metric's second half, as in round 1
        if args.cpu_budget > 0:
            # rank 0 only, whatever the world size (after the timed region and after the process group is gone)
            base = cpu_baseline(host_x[0], n_fft, hop, consts["weight"], consts["lut"], args.cpu_budget, with_legs=bool(legs) or stub)
            side = base.pop("_side")
            result["cpu_baseline"] = base
            for leg_name, side_name in (("configs2_bank_iir_time_parallel", "octave_iir_bpo3"), ("configs2_bank_iir_sequential", "octave_iir_bpo3"),
                                        ("configs2_bank_fir_overlap_add", "octave_ola_bpo3"), ("configs4_bank_iir_time_parallel", "octave_iir_bpo24"),
                                        ("configs4_bank_fir_overlap_add", "octave_ola_bpo24"), ("configs4_gcc_phat", "gcc_phat"),
                                        ("configs4_gcc_phat_1024_pairs", "gcc_phat")):
                if leg_name in legs and side_name in side:
                    legs[leg_name]["cpu_baseline"] = side[side_name]
            if stub:
                result["cpu_baseline_legs"] = side
        if args.full_json:
            Path(args.full_json).write_text(json.dumps(result, indent=1) + "\n")
        line = json.dumps(compact_line(result), separators=(",", ":"))
        if len(line) > LINE_BUDGET:
            print(f"bench.py: the JSON line is {len(line)} characters (budget {LINE_BUDGET}): the driver's tail may cut it", file=sys.stderr)
        print(line, flush=True)
        gates = [result.get("parity", {}).get("gate", {}).get("pass", True)]
        gates += [lg.get("parity", {}).get("gate", {}).get("pass", True) for lg in legs.values() if isinstance(lg, dict)]
        if slab_check is not None:
            gates.append(slab_check)
        if not all(gates):
            print("bench.py: a parity gate failed (see `parity` / legs[*].parity in the line above)", file=sys.stderr)
            exit_code = 3
    if ranks_seen != list(range(world)):
        print(f"bench.py: ranks_seen = {ranks_seen}, expected 0..{world - 1}", file=sys.stderr)
        exit_code = 4

    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
