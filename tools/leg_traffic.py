#!/usr/bin/env python
"""HBM traffic of every bench leg's kernels from rocprofv3 PMC passes -> profiles/r06_leg_traffic.json (bench.py quotes it as each
leg's roofline.traffic when the digest of the leg's kernel sources matches the sources it runs; tests/test_evidence_fresh.py).

    python tools/leg_traffic.py run <leg> [--calls 4] [--out <dir>/<leg>]   one leg's workload, CALLS identical calls (under rocprofv3 --pmc)
    python tools/leg_traffic.py collect <dir>             <dir>/<leg>/<pass>/**/*counter_collection.csv -> the json
    python tools/leg_traffic.py legs                      the leg names

tools/gpu_leg_traffic.sh drives the passes (FETCH_SIZE, WRITE_SIZE and the TCC_EA0 request counters in separate runs, kernel-trace
only).  Corrections as MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE (KB) tallies 128-byte read requests at 64 bytes ->
doubled, cross-checked against TCC_EA0_RDREQ (32-byte requests counted apart); WRITE_SIZE (KB) as is, cross-checked against
TCC_EA0_WRREQ x 64 B.  A kernel counts towards a leg when it was launched in every call (its dispatch count is a multiple of the
number of calls): one-off table kernels of handle creation are listed apart."""
import collections
import csv
import glob
import json
import os
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
CSRC = ROOT / "friture_amd" / "csrc"

import bench  # noqa: E402  (LEG_SOURCES, sources_digest: the digest bench.py checks before quoting a figure)

# leg of this tool -> the bench.py leg it serves
LEGS = {"gcc1024": "configs4_gcc_phat_1024_pairs", "gcc100": "configs4_gcc_phat", "iir3": "configs2_bank_iir_time_parallel",
        "iir24": "configs4_bank_iir_time_parallel", "ola3": "configs2_bank_fir_overlap_add", "ola24": "configs4_bank_fir_overlap_add",
        "stft16384_psd": "configs3_stft16384_psd", "stft16384_image": "configs3_stft16384_image",
        "stft16384_hop4096_psd": "configs3_stft16384_hop4096_psd", "stft16384_hop4096_image": "configs3_stft16384_hop4096_image",
        "f64_psd": "configs1_f64_psd", "f64_image": "configs1_f64_image"}


def workload(leg):
    """(callable of one call, algorithmic HBM bytes per call) — the same shapes as bench.py's legs."""
    import torch

    from friture_amd import _lib, filter_design, palette, tables
    dev = torch.device("cuda", 0)
    _lib.init(0)
    if leg.startswith("gcc"):
        from friture_amd.signal.correlation import GccPhat
        pairs, L = int(leg[3:]), 24000
        rng = np.random.default_rng(4242)
        d0 = 0.25 * rng.standard_normal((pairs, L))
        d1 = np.roll(d0, 37, axis=1) + 0.025 * rng.standard_normal((pairs, L))
        g = GccPhat(L, pairs)
        a0, a1 = torch.from_numpy(d0).to(dev), torch.from_numpy(d1).to(dev)
        return (lambda: g.correlate(a0, a1)), pairs * 24 * L
    if leg.startswith(("iir", "ola")):
        from friture_amd.filter import FirBank, IirBank
        bpo = int(leg[3:])
        ch, log2n = (8, 22) if bpo == 3 else (8, 20)
        n = 1 << log2n
        t = filter_design.load_tables()
        x = torch.from_numpy(np.stack([bench.synth_channel(1000 + c, n) for c in range(ch)])).to(dev)
        decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
        alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
        out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device=dev)
        if leg.startswith("iir"):
            bank = IirBank(t["bdec"], t["adec"], list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"]), ch)
            bank.set_chunk(1024 if bpo <= 3 else 512)
        else:
            bank = FirBank(bpo, ch, t)
        return (lambda: bank.energies(x, 1024, alphas, out=out)), ch * (n // 1024) * (4096 + 4 * 9 * bpo)
    from friture_amd.stft import StftEngine
    lut = palette.cmr_lut()
    if leg.startswith("stft16384"):
        n_fft, ch, T = 16384, 32, 1 << 20
        hop = 4096 if "hop4096" in leg else 8192
        kind = 3 if leg.endswith("image") else 0
        weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
        x = torch.from_numpy(np.stack([bench.synth_channel(5000 + c, T) for c in range(ch)])).to(dev)
        eng = StftEngine(n_fft, hop, ch, 32)
        eng.set_epilogue(weight, -140.0, 0.0, lut)
        F = eng.frames_for(T)
        out = torch.empty((ch, F, n_fft // 2 + 1), dtype=torch.int32 if kind == 3 else torch.float32, device=dev)
        return (lambda: eng.run(kind, x, out)), ch * F * (4 * hop + 4 * (n_fft // 2 + 1))
    if leg.startswith("f64"):
        n_fft, hop, T = 1024, 512, 1 << 25
        kind, ob = (3, 4) if leg.endswith("image") else (0, 8)
        weight = tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0]
        x = torch.from_numpy(np.stack([bench.synth_channel(7000, T).astype(np.float64)])).to(dev)
        eng = StftEngine(n_fft, hop, 1, 64)
        eng.set_epilogue(weight, -140.0, 0.0, lut)
        F = eng.frames_for(T)
        nb = n_fft // 2 + 1
        slab = torch.empty((F * nb,), dtype=torch.int32 if kind == 3 else torch.float64, device=dev)
        rows, nyq = slab[:F * (nb - 1)].view(1, F, nb - 1), slab[F * (nb - 1):].view(1, F)
        return (lambda: eng.run_split(kind, x, rows, nyq)), F * (8 * hop + ob * nb)
    raise SystemExit(f"unknown leg {leg}")


def run(leg, calls, out_dir=None):
    import torch
    fn, alg = workload(leg)
    if out_dir:                                     # the sizes next to the counters, for collect
        Path(out_dir).mkdir(parents=True, exist_ok=True)
        (Path(out_dir) / "calls").write_text(str(calls))
        (Path(out_dir) / "algorithmic_bytes").write_text(str(alg))
    torch.cuda.synchronize()
    for _ in range(calls):
        fn()
        torch.cuda.synchronize()
    print(f"leg {leg}: {calls} calls")


def collect(root, out_path=None):
    rec = {}
    for leg, bench_leg in LEGS.items():
        bench_legs, sources = [bench_leg], bench.LEG_SOURCES[bench_leg]
        files = glob.glob(os.path.join(root, leg, "*", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        calls_file = Path(root) / leg / "calls"
        calls = int(calls_file.read_text()) if calls_file.exists() else 4
        acc = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> values
        for f in files:
            for r in csv.DictReader(open(f)):
                name = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        kernels, oneoff = {}, {}
        for name, ctr in acc.items():
            n = len(ctr.get("FETCH_SIZE", ctr.get("WRITE_SIZE", [])))
            entry = {"launches_per_call": n / calls,
                     "read_bytes_per_call": 2.0 * 1024.0 * sum(ctr.get("FETCH_SIZE", [])) / calls,
                     "write_bytes_per_call": 1024.0 * sum(ctr.get("WRITE_SIZE", [])) / calls}
            if "TCC_EA0_RDREQ_sum" in ctr:
                rd, rd32 = sum(ctr["TCC_EA0_RDREQ_sum"]) / calls, sum(ctr.get("TCC_EA0_RDREQ_32B_sum", [0.0])) / calls
                entry["check_read_requests_bytes"] = (rd - rd32) * 128.0 + rd32 * 32.0
                wr, wr64 = sum(ctr["TCC_EA0_WRREQ_sum"]) / calls, sum(ctr.get("TCC_EA0_WRREQ_64B_sum", [0.0])) / calls
                entry["check_write_requests_bytes"] = wr64 * 64.0 + (wr - wr64) * 32.0
            (kernels if n and n % calls == 0 else oneoff)[name] = entry
        if not kernels:
            continue
        total = sum(k["read_bytes_per_call"] + k["write_bytes_per_call"] for k in kernels.values())
        dominant = max(kernels, key=lambda k: kernels[k]["read_bytes_per_call"] + kernels[k]["write_bytes_per_call"])
        alg_file = Path(root) / leg / "algorithmic_bytes"
        alg = float(alg_file.read_text()) if alg_file.exists() else None
        rec[leg] = {"bench_legs": bench_legs, "kernel_sources": bench.sources_digest(sources), "sources": sources, "calls": calls,
                    "hbm_bytes_per_call": total, "read_bytes_per_call": sum(k["read_bytes_per_call"] for k in kernels.values()),
                    "write_bytes_per_call": sum(k["write_bytes_per_call"] for k in kernels.values()),
                    "algorithmic_bytes_per_call": alg, "ratio": total / alg if alg else None,
                    "dominant_kernel": dominant, "kernels": kernels, "one_off_kernels": sorted(oneoff)}
    target = Path(out_path or ROOT / "profiles" / "r06_leg_traffic.json")
    if target.exists():                     # a session that re-measures some legs keeps the others' records
        try:
            old = json.loads(target.read_text()).get("legs", {})
            rec = {**{k: v for k, v in old.items() if k not in rec}, **rec}
            rec = {k: rec[k] for k in LEGS if k in rec}
        except Exception:
            pass
    out = {"method": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python tools/leg_traffic.py run <leg>; all dispatches "
                     "of the kernels launched in every call, summed per call; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md "
                     "(128-byte requests tallied at 64 B), WRITE_SIZE (KB) as is; check_* from TCC_EA0_RDREQ / WRREQ request counts",
           "legs": rec}
    target.write_text(json.dumps(out, indent=1) + "\n")
    for leg, r in rec.items():
        print(f"{leg:26s} {r['hbm_bytes_per_call'] / 1e6:10.2f} MB/call  algorithmic {(r['algorithmic_bytes_per_call'] or 0) / 1e6:9.2f} MB  "
              f"x{r['ratio'] or 0:.3f}  dominant {r['dominant_kernel'][:60]}")


def main():
    if len(sys.argv) < 2 or sys.argv[1] == "legs":
        print(" ".join(LEGS))
        return
    if sys.argv[1] == "run":
        leg = sys.argv[2]
        calls = int(sys.argv[sys.argv.index("--calls") + 1]) if "--calls" in sys.argv else 4
        for i, a in enumerate(sys.argv):             # --set-option name=value: force a product code path (A/B records)
            if a == "--set-option":
                from friture_amd import _lib
                name, value = sys.argv[i + 1].split("=")
                _lib.set_option(name, int(value))
        run(leg, calls, sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None)
    elif sys.argv[1] == "collect":
        collect(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
