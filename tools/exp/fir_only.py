"""FIR overlap-add bank only (for profiling): python tools/exp/fir_only.py [ch bpo log2n steps]"""
import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from friture_amd import _lib, filter_design
from friture_amd.filter import FirBank
import os
if os.environ.get('FRT_LIB_VARIANT'):      # A/B runs: a variant library built by tools/exp/build_variant.sh
    _lib.LIB_PATH = Path(__file__).resolve().parents[1] / 'variants' / os.environ['FRT_LIB_VARIANT'] / 'libfriture_hip.so'
_lib.init(0)
ch, bpo, log2n, steps = (int(v) for v in (sys.argv[1:5] + ["8", "3", "22", "10"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
t = filter_design.load_tables()
n = 1 << log2n
x = 0.25 * torch.randn((ch, n), device=dev, dtype=torch.float32)
decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
bank = FirBank(bpo, ch, t)
for _ in range(3):
    bank.energies(x, 1024, alphas, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bank.energies(x, 1024, alphas, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"ch": ch, "bpo": bpo, "log2n": log2n, "ms": dt * 1e3, "digest": float(out.double().mean().item())}))
