#!/usr/bin/env python
"""HBM bytes per stft_kernel launch from the rocprofv3 PMC passes of `bench.py` (tools/gpu_session.sh) ->
profiles/pmc_traffic.json, stamped with the digest of the kernel sources it was measured on (bench.py quotes the
figure as roofline.traffic only when that digest matches the sources it runs).

    python tools/pmc_traffic.py gpurun_out/prof/pmc_bench  [kernel-substring = stft_kernel]

Corrections as MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE (KB) tallies the 128-byte read requests of a
wide coalesced stream at 64 bytes -> doubled; cross-checked against TCC_EA0_RDREQ x 128 B.  WRITE_SIZE (KB) as is,
cross-checked against TCC_EA0_WRREQ x 64 B."""
import collections
import csv
import glob
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    root = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else "stft_kernel"
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the bench command also launches the kernel once on a 4096-frame sample (its parity report): keep the full-size launches
    full = {k: [x for x in v if x > 0.5 * max(v)] for k, v in acc.items()}
    raw = {k: sum(v) / len(v) for k, v in full.items()}
    acc = full
    if "FETCH_SIZE" not in raw or "WRITE_SIZE" not in raw:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE rows for '{needle}' under {root}: {sorted(raw)}")
    import bench
    read_b, write_b = 2.0 * raw["FETCH_SIZE"] * 1024.0, raw["WRITE_SIZE"] * 1024.0
    rec = {"n_fft": 1024, "hop": 512, "frames": 131071, "kernel": f"{needle} (colour and PSD kinds of bench.py, split rows)",
           "kernel_sources": bench.kernel_source_digest(),
           "hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b, "raw": raw,
           "check": {"TCC_EA0_RDREQ*128B": raw.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0, "TCC_EA0_WRREQ*64B": raw.get("TCC_EA0_WRREQ_sum", 0.0) * 64.0},
           "dispatches": {k: len(v) for k, v in acc.items()},
           "method": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python bench.py --steps 5 --warmup 2 --cpu-budget 0 "
                     "--prewarm-ms 0 --no-legs; mean over the kernel's dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "algorithmic_bytes_per_launch": 131071 * 4100,
           "kernel_sources_note": "digest of the code of stft_wave.h + fft_core.h + stft_launch of stft.hip (comments and blank lines removed) at the "
                                  "measured revision; bench.kernel_source_digest()"}
    (ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(rec, indent=1))
    print(json.dumps({k: rec[k] for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes", "algorithmic_bytes_per_launch", "kernel_sources")}))


if __name__ == "__main__":
    main()
