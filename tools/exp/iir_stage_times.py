#!/usr/bin/env python
"""IirBank.energies (time-parallel mode) a few times for one shape (run under rocprofv3 --kernel-trace), or, with --parse DIR, the
launch sequence of one call: kernel, grid, duration."""
import csv, glob, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-28:],
                         int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
    rows.sort()
    names = [r[2] for r in rows]
    # one call = from an energy_finish / energy_scan kernel to the next: take the last complete call
    ends = [i for i, n in enumerate(names) if "energy_finish" in n or "energy_scan" in n]
    if len(ends) < 2:
        print("no complete call found"); sys.exit(0)
    a, b = ends[-2] + 1, ends[-1] + 1
    t0 = rows[a][0]
    tot = 0.0
    for s, e, n, gx, gy, gz in rows[a:b]:
        print(f"{(s - t0) / 1e3:8.1f} us  {n:30s} grid {gx:6d} x {gy:3d} x {gz:3d}  {(e - s) / 1e3:7.1f} us")
        tot += (e - s) / 1e3
    print(f"{b - a} launches, kernel time {tot:.1f} us, span {(rows[b - 1][1] - t0) / 1e3:.1f} us")
    sys.exit(0)
import numpy as np, torch
from friture_amd import _lib, filter_design
from friture_amd.filter import IirBank
ch, bpo, log2n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 3, 22)
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
import os
if os.environ.get("FRT_LIB_VARIANT"):      # A/B runs: a variant library built by tools/exp/build_variant.sh
    _lib.LIB_PATH = ROOT / "tools" / "variants" / os.environ["FRT_LIB_VARIANT"] / "libfriture_hip.so"
_lib.init(0)
for kv in os.environ.get("FRT_OPTIONS", "").split():      # product options for A/B runs: FRT_OPTIONS="iir_lane_columns=0 iir_lookback=0"
    name, value = kv.split("=")
    _lib.set_option(name, int(value))
dev = torch.device("cuda", 0)
t = filter_design.load_tables()
n = 1 << log2n
x = 0.25 * torch.randn((ch, n), device=dev, dtype=torch.float32)
decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
bank = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
bank.set_chunk(chunk)
for _ in range(6):
    bank.energies(x, 1024, alphas, out=out)
torch.cuda.synchronize()
