"""Timing of the octave-bank energies path (BASELINE configs[2]: 8 ch, bpo = 3, 48 kHz) on one GPU.

    python tools/bench_octbank.py [--channels 8] [--bpo 3] [--log2-samples 22] [--chunk 16384] [--iters 5]

Unit: octave-bands/s = channels * blocks * 9*bpo / time (one unit = one band's filtered output +
smoothed energy for one 1024-sample block of one channel).
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--bpo", type=int, default=3)
    ap.add_argument("--log2-samples", type=int, default=22)
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cpu-blocks", type=int, default=0, help="time the oracle on this many blocks of channel 0")
    args = ap.parse_args()
    import torch
    from friture_amd import _lib, filter_design
    from friture_amd.filter import IirBank
    import os
    if os.environ.get('FRT_LIB_VARIANT'):      # A/B runs: a variant library built by tools/exp/build_variant.sh
        _lib.LIB_PATH = Path(__file__).resolve().parent / 'variants' / os.environ['FRT_LIB_VARIANT'] / 'libfriture_hip.so'
    _lib.init(0)
    t = filter_design.load_tables()
    bank = IirBank(t["bdec"], t["adec"], list(t[f"boct_{args.bpo}"]), list(t[f"aoct_{args.bpo}"]), args.channels)
    bank.set_chunk(args.chunk)
    n = 1 << args.log2_samples
    rng = np.random.default_rng(42)
    x = torch.from_numpy((0.25 * rng.standard_normal((args.channels, n))).astype(np.float32)).cuda()
    decs = [2 ** j for j in range(9)[::-1] for _ in range(args.bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
    out = torch.empty((args.channels, n // 1024, 9 * args.bpo), dtype=torch.float32, device="cuda")
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.25:                 # GPU clock ramp: reach the sustained clocks first
        bank.energies(x, 1024, alphas, out=out)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        bank.energies(x, 1024, alphas, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    units = args.channels * (n // 1024) * 9 * args.bpo
    bytes_alg = args.channels * (n // 1024) * (4096 + 4 * 9 * args.bpo)
    res = {"octave_bands_per_s": units / dt, "ms": dt * 1e3, "channels": args.channels, "bpo": args.bpo, "samples": n,
           "chunk": args.chunk, "samples_per_s": args.channels * n / dt, "algorithmic_GBps": bytes_alg / dt / 1e9}
    if args.cpu_blocks:
        from oracle import dsp
        boct, aoct = list(t[f"boct_{args.bpo}"]), list(t[f"aoct_{args.bpo}"])
        zs = dsp.iir_bank_filtic(t["bdec"], t["adec"], boct, aoct)
        al, kern = dsp.band_smoothing_setup(args.bpo, 1.0)
        prev = [0.0] * (9 * args.bpo)
        xs = x[0, :1024 * args.cpu_blocks].cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        for b in range(args.cpu_blocks):
            y, _, zs = dsp.iir_bank(t["bdec"], t["adec"], boct, aoct, xs[b * 1024:(b + 1) * 1024], zs)
            prev = dsp.band_energies(y, kern, al, prev)
        res["cpu_octave_bands_per_s"] = args.cpu_blocks * 9 * args.bpo / (time.perf_counter() - t0)
        res["cpu_kind"] = "oracle (C DF2T loop + numpy), 1 core"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
