"""Extract the reference's published filter-design numbers into friture_amd/data/octave_filters.npz.

The octave-bank coefficients are part of the reference's contract: every parity target is defined
with exactly these numbers (friture/generated_filters.py JSON, friture/data/generated_fft.npz).
friture_amd/filter_design.py re-derives the same designs, but scipy's elliptic design routines
changed between the version upstream used and the one installed here, so the re-derived values
differ by 1e-5 .. 3e-3 (relative) — far more than the 1e-5 parity tolerance on band energies.
The shipped table therefore holds the reference's numbers verbatim (values only, no code), and
tests/test_filter_tables.py keeps the re-derivation honest.

Run in the build container only (needs /root/reference):  python tools/extract_reference_tables.py
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parents[1] / "friture_amd" / "data" / "octave_filters.npz"

text = (REF / "friture" / "generated_filters.py").read_text()
params = json.loads(re.search(r'JSON_PARAMS = """(.*?)"""', text, re.S).group(1))
fft = np.load(REF / "friture" / "data" / "generated_fft.npz")

out = {
    "bdec": np.asarray(params["dec"][0], float),
    "adec": np.asarray(params["dec"][1], float),
    "bdec_fir": np.asarray(fft["bdec_fir"], float),
}
sizes = None
for bpo in (1, 3, 6, 12, 24):
    boct, aoct, fi, flow, fhigh = params[str(bpo)]
    out[f"boct_{bpo}"] = np.asarray(boct, float)
    out[f"aoct_{bpo}"] = np.asarray(aoct, float)
    out[f"boct_fir_{bpo}"] = np.asarray(fft[f"{bpo}_boct_fir"], float)
    s = np.asarray(fft[f"{bpo}_fft_sizes"], np.int64)
    assert sizes is None or (s == sizes).all()
    sizes = s
out["fft_sizes"] = sizes
OUT.parent.mkdir(parents=True, exist_ok=True)
np.savez_compressed(OUT, **out)
print("wrote", OUT, {k: v.shape for k, v in out.items()})
