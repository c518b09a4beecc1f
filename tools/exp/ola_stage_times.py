#!/usr/bin/env python
"""FirBank.energies a few times for one shape (run under rocprofv3 --kernel-trace), or, with --parse DIR, the per-launch durations
of the batched overlap-add kernel grouped by grid size (one group per octave stage)."""
import csv, glob, json, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = defaultdict(list)
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ola_" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0][-24:], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
                rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = 0.0
    for k, v in sorted(rows.items(), key=lambda kv: -kv[0][1]):
        v = sorted(v)
        med = v[len(v) // 2]
        tot += med
        print(f"{k[0]:24s} grid {k[1]:6d} x {k[2]:2d} x {k[3]:3d}  launches {len(v):3d}  median {med:8.1f} us  min {v[0]:8.1f}")
    print(f"sum of medians {tot:.1f} us")
    sys.exit(0)
import numpy as np, torch
from friture_amd import _lib, filter_design
from friture_amd.filter import FirBank
ch, bpo, log2n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 3, 22)
_lib.init(0)
dev = torch.device("cuda", 0)
t = filter_design.load_tables()
n = 1 << log2n
x = 0.25 * torch.randn((ch, n), device=dev, dtype=torch.float32)
decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
bank = FirBank(bpo, ch, t)
for _ in range(8):
    bank.energies(x, 1024, alphas, out=out)
torch.cuda.synchronize()
