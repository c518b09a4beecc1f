#!/usr/bin/env python
"""Batched FFT overlap-add bank (FirBank) against the exact IIR bank: octave-bands/s for BASELINE configs[2] / [4] shapes."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
from friture_amd import _lib, filter_design
from friture_amd.filter import FirBank, IirBank

import os
if os.environ.get("FRT_LIB_VARIANT"):      # A/B runs: a variant library built by tools/exp/build_variant.sh
    _lib.LIB_PATH = ROOT / "tools" / "variants" / os.environ["FRT_LIB_VARIANT"] / "libfriture_hip.so"
_lib.init(0)
dev = torch.device("cuda", 0)
t = filter_design.load_tables()
for ch, bpo, log2n in ((8, 3, 22), (64, 24, 20), (8, 24, 20)):
    n = 1 << log2n
    x = (0.25 * torch.randn((ch, n), device=dev, dtype=torch.float32))
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
    out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
    banks = {"fir": FirBank(bpo, ch, t)}
    iir = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
    iir.set_chunk(1024)
    banks["iir"] = iir
    res = {}
    for name, bank in banks.items():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            bank.energies(x, 1024, alphas, out=out)
            torch.cuda.synchronize()
        steps = 20
        t0 = time.perf_counter()
        for _ in range(steps):
            bank.energies(x, 1024, alphas, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[name] = {"ms": dt * 1e3, "octave_bands_per_s": ch * (n // 1024) * 9 * bpo / dt, "digest": float(out.double().mean().item())}
    print(json.dumps({"channels": ch, "bpo": bpo, "log2n": log2n, **res}))
