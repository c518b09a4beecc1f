"""Where the fixed host time of bench.py's timed region goes: 20 launches of the headline kernel bracketed as
bench.timed() brackets them, every host step stamped.  usage: python tools/exp/host_fixed.py [--spin] [--steps 20] [--reps 30]

--spin: hipSetDeviceFlags(hipDeviceScheduleSpin) before the runtime is initialised (the host thread spins on the completion signal
instead of sleeping on an interrupt)."""
import argparse
import ctypes
import json
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
ap = argparse.ArgumentParser()
ap.add_argument("--spin", action="store_true")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
flag_rc = None
if args.spin:
    hip = ctypes.CDLL("libamdhip64.so")
    flag_rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))          # hipDeviceScheduleSpin
import torch

import bench
from friture_amd import _lib, palette, tables
from friture_amd.stft import StftEngine

dev = torch.device("cuda", 0)
_lib.init(0)
n_fft, hop, T = 1024, 512, 1 << 26
xs = [torch.from_numpy(bench.synth_channel(100000 * b, T)[None]).to(dev) for b in range(3)]
eng = StftEngine(n_fft, hop, 1, 32)
eng.set_epilogue(tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0], -140.0, 0.0, palette.cmr_lut())
F = eng.frames_for(T)
slabs = [torch.empty((F * 513,), dtype=torch.int32, device=dev) for _ in range(3)]
rows = [s[:F * 512].view(1, F, 512) for s in slabs]
nyqs = [s[F * 512:].view(1, F) for s in slabs]


def step(k):
    b = k % 3
    eng.run_split(3, xs[b], rows[b], nyqs[b])


# the same launch with everything the call needs extracted once (what a prepared call would cost)
lib = eng._lib
nf = ctypes.c_int64(0)
prepared = [(ctypes.c_void_p(xs[b].data_ptr()), ctypes.c_void_p(rows[b].data_ptr()), ctypes.c_void_p(nyqs[b].data_ptr())) for b in range(3)]
lib.frt_stft_set_stream(eng._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
nfref = ctypes.byref(nf)


def step_prepared(k):
    x, r, n = prepared[k % 3]
    lib.frt_stft_run_split(eng._h, 3, x, T, T, r, n, nfref)


t_end = time.perf_counter() + 0.4
k = 0
while time.perf_counter() < t_end:
    step(k)
    k += 1
torch.cuda.synchronize()
out = {"spin": args.spin, "hipSetDeviceFlags_rc": flag_rc, "steps": args.steps}
for name, fn in (("run_split", step), ("prepared", step_prepared)):
    rec = {k: [] for k in ("record0", "first_call", "other_calls", "record1", "sync", "wall", "gpu", "fixed")}
    for rep in range(args.reps):
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        t1 = time.perf_counter()
        fn(0)
        t2 = time.perf_counter()
        for k in range(1, args.steps):
            fn(k)
        t3 = time.perf_counter()
        ev1.record()
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        gpu = ev0.elapsed_time(ev1) * 1e3
        for key, v in (("record0", t1 - t0), ("first_call", t2 - t1), ("other_calls", t3 - t2), ("record1", t4 - t3), ("sync", t5 - t4),
                       ("wall", t5 - t0)):
            rec[key].append(v * 1e6)
        rec["gpu"].append(gpu)
        rec["fixed"].append((t5 - t0) * 1e6 - gpu)
    out[name] = {k: round(statistics.median(v), 1) for k, v in rec.items()}
    out[name]["fixed_min"] = round(min(rec["fixed"]), 1)
    out[name]["wall_per_step_us"] = round(out[name]["wall"] / args.steps, 3)
    out[name]["gpu_per_step_us"] = round(out[name]["gpu"] / args.steps, 3)
print(json.dumps(out))
