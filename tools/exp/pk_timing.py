"""Per-interval cycle counts of the packed N = 16384 instance from a -DFRT_PK_TIMING=1 variant library (wrong spectra by
design): FRT_LIB_VARIANT=<name> python tools/exp/pk_timing.py [hop] [kind]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from friture_amd import _lib
if os.environ.get("FRT_LIB_VARIANT"):
    _lib.LIB_PATH = Path(__file__).resolve().parents[1] / "variants" / os.environ["FRT_LIB_VARIANT"] / "libfriture_hip.so"
from friture_amd.stft import StftEngine
from friture_amd import tables, palette
_lib.init(0)
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, C, T = 16384, 32, 1 << 20
x = 0.25 * torch.randn((C, T), device="cuda", dtype=torch.float32)
if os.environ.get("PK_ZERO_INPUT"):
    x.zero_()
e = StftEngine(N, hop, C, 32)
e.set_epilogue(tables.weighting_db(tables.rfft_frequencies(N), 1e-50)[0], -140.0, 0.0, palette.cmr_lut())
F = e.frames_for(T)
out = torch.empty((C, F, N // 2 + 1), dtype=torch.int32 if kind == 3 else torch.float32, device="cuda")
for _ in range(20):
    e.run(kind, x, out)
torch.cuda.synchronize()
o = out.view(torch.float32).cpu().numpy()
# rows that start a run carry 64 floats of timing; find them: run length from the first channel
rows = [f for f in range(F) if np.all(o[0, f, :64] > 1.0) and np.all(o[0, f, :64] < 1e7)]
run = rows[1] - rows[0] if len(rows) > 1 else F
acc = np.array([o[c, f, :64].reshape(8, 8) for c in range(C) for f in rows])          # [runs][wave][interval]
ghz = np.array([o[c, f, 64:72] for c in range(C) for f in rows])
print(f"shader clock during the runs (s_memtime / s_memrealtime): mean {ghz.mean():.3f} GHz, min {ghz.min():.3f}, max {ghz.max():.3f}")
names = ["copy wait + ring reads + window", "DFT16 + twiddles (+ copy issue)", "barrier A", "transpose writes", "barrier B",
         "sub-transforms (3 passes, 2 rounds)", "barrier C", "unpack + stores"]
mean = acc.mean(axis=0)
print(f"hop {hop} kind {kind}: runs of {run} frames, {len(acc)} runs; cycles per frame (s_memtime ticks), mean over runs")
print("interval".ljust(40), " ".join(f"wave{w}".rjust(7) for w in range(8)), "   mean")
for i, n in enumerate(names):
    print(n.ljust(40), " ".join(f"{mean[w, i]:7.0f}" for w in range(8)), f"{mean[:, i].mean():7.0f}")
print("total".ljust(40), " ".join(f"{mean[w].sum():7.0f}" for w in range(8)), f"{mean.sum(axis=1).mean():7.0f}")
