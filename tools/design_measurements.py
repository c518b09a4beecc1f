#!/usr/bin/env python
"""Rewrites DESIGN.md §6's table from profiles/r06_bench_full.json (the --full-json record of the round's evidence session), between
the markers <!-- r06-table-begin --> / <!-- r06-table-end -->, so that the document's numbers are the committed record's."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
d = json.loads((ROOT / "profiles" / "r06_bench_full.json").read_text())
packed = json.loads((ROOT / "profiles" / "r06_bench_packed.json").read_text())
traffic = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())["hbm_bytes_per_launch"]      # the PMC passes of the same session
L = d["legs"]


def e(x, digits=3):
    m, p = f"{x:.{digits - 1}e}".split("e")
    return f"{m}x10^{int(p)}"


def cpu(leg):
    b = leg.get("cpu_baseline")
    if not b:
        return ""
    return f"{e(b['value'], 2)} / {e(b['all_cores']['value'], 2)}" if "all_cores" in b else e(b["value"], 2)


r = d["roofline"]
rows = [
    ("**headline** configs[1], float32, colour, split rows", f"**{e(d['value'], 4)} spectra/s**, {d['ms_per_step']:.4f} ms",
     f"**{r['frac']:.3f}** of 8 TB/s (kernel {r['kernel_ms']:.4f} ms; packed rows, same session: {packed['roofline']['frac']:.3f}); traffic "
     f"{traffic / 1e6:.1f} MB vs {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic", "0.601 (1.130x10^9)",
     f"{e(d['cpu_baseline']['value'], 2)} / {e(d['cpu_baseline']['all_cores']['value'], 2)} spectra/s"),
    ("same batches, PSD kind", e(d["psd_output"]["spectra_per_s"], 4), f"{d['psd_output']['frac_of_hbm_peak']:.3f}", "0.643", ""),
    ("`configs1_f64_psd` / `_image` (split rows)", f"{e(L['configs1_f64_psd']['value'])} / {e(L['configs1_f64_image']['value'])}",
     f"{L['configs1_f64_psd']['roofline']['frac']:.3f} / {L['configs1_f64_image']['roofline']['frac']:.3f}", "0.663 / 0.518", ""),
    ("`configs2_bank_iir_time_parallel` (= `octave_bands`)", f"**{e(L['configs2_bank_iir_time_parallel']['value'])} octave-bands/s**, "
     f"{L['configs2_bank_iir_time_parallel']['ms_per_step']:.3f} ms", f"{L['configs2_bank_iir_time_parallel']['roofline']['f64_frac']:.3f} of the f64 vector peak (SURVEY's flop model)",
     "0.618 ms", cpu(L["configs2_bank_iir_time_parallel"])),
    ("`configs2_bank_iir_sequential` (bit-exact)", f"{e(L['configs2_bank_iir_sequential']['value'])}, {L['configs2_bank_iir_sequential']['ms_per_step']:.2f} ms (8 ch x 2^16)",
     "serial in time", "2.19x10^6", ""),
    ("`configs2_bank_fir_overlap_add`", f"**{e(L['configs2_bank_fir_overlap_add']['value'])}**, {L['configs2_bank_fir_overlap_add']['ms_per_step']:.3f} ms",
     f"{L['configs2_bank_fir_overlap_add']['roofline']['f64_frac']:.3f} of the f64 peak", "0.793 ms", cpu(L["configs2_bank_fir_overlap_add"])),
    ("`configs3_stft16384_psd` / `_image`", f"{e(L['configs3_stft16384_psd']['value'])} / {e(L['configs3_stft16384_image']['value'])} spectra/s",
     f"**{L['configs3_stft16384_psd']['roofline']['frac']:.3f} / {L['configs3_stft16384_image']['roofline']['frac']:.3f}**", "0.465 / 0.388", ""),
    ("`configs3_stft16384_hop4096_psd` / `_image`", f"{e(L['configs3_stft16384_hop4096_psd']['value'])} / {e(L['configs3_stft16384_hop4096_image']['value'])}",
     f"**{L['configs3_stft16384_hop4096_psd']['roofline']['frac']:.3f} / {L['configs3_stft16384_hop4096_image']['roofline']['frac']:.3f}**", "0.427 / 0.355", ""),
    ("`configs4_gcc_phat` / `_1024_pairs`", f"{e(L['configs4_gcc_phat']['value'])} / **{e(L['configs4_gcc_phat_1024_pairs']['value'])} windows/s**",
     f"{L['configs4_gcc_phat']['roofline']['frac']:.3f} / {L['configs4_gcc_phat_1024_pairs']['roofline']['frac']:.3f} of HBM", "1.16 / 2.05x10^6 (0.083 / 0.148)", cpu(L["configs4_gcc_phat"]) + " windows/s"),
    ("`configs4_bank_iir_time_parallel` (chunk 512) / `_fir` (216 bands)",
     f"{e(L['configs4_bank_iir_time_parallel']['value'])} ({L['configs4_bank_iir_time_parallel']['ms_per_step']:.3f} ms) / "
     f"**{e(L['configs4_bank_fir_overlap_add']['value'])} ({L['configs4_bank_fir_overlap_add']['ms_per_step']:.3f} ms)**",
     f"{L['configs4_bank_iir_time_parallel']['roofline']['f64_frac']:.3f} / {L['configs4_bank_fir_overlap_add']['roofline']['f64_frac']:.3f} of the f64 peak",
     "0.691 / 1.092 ms", cpu(L["configs4_bank_iir_time_parallel"])),
]
lt = json.loads((ROOT / "profiles" / "r06_leg_traffic.json").read_text())["legs"]
tr = {name: rec for rec in lt.values() for name in rec["bench_legs"]}


def x(*names):
    return "; traffic " + " / ".join(f"x{tr[n]['ratio']:.2f}" for n in names) + " of the algorithmic bytes"


extra = [None, None, x("configs1_f64_psd", "configs1_f64_image"), x("configs2_bank_iir_time_parallel"), None, x("configs2_bank_fir_overlap_add"),
         x("configs3_stft16384_psd", "configs3_stft16384_image"), x("configs3_stft16384_hop4096_psd", "configs3_stft16384_hop4096_image"),
         x("configs4_gcc_phat", "configs4_gcc_phat_1024_pairs"), x("configs4_bank_iir_time_parallel", "configs4_bank_fir_overlap_add")]
rows = [(r0[0], r0[1], r0[2] + (extra[i] or ""), r0[3], r0[4]) for i, r0 in enumerate(rows)]
table = ["| line / leg | rate | roofline | round 5 (driver) | CPU baseline (oracle `port`, same run: 1 core / 64 processes) |", "|---|---|---|---|---|"]
table += ["| " + " | ".join(row) + " |" for row in rows]
p = d["parity"]
tail = (f"\nParity of the timed batch (split rows reassembled): `gate.pass` ({p['epilogue_mismatch_outside_edge']} epilogue mismatches, {p['pixels_mismatched']} of "
        f"{p['pixels_checked'] / 1e6:.1f} M pixels differ from the float64 image, {p['mismatch_unaccounted']} unaccounted for; float32 PSD within {p['psd_rel_max']:.1e}).")
text = (ROOT / "DESIGN.md").read_text()
new = "<!-- r06-table-begin -->\n" + "\n".join(table) + "\n" + tail + "\n<!-- r06-table-end -->"
text, n = re.subn(r"<!-- r06-table-begin -->.*?<!-- r06-table-end -->", lambda m: new, text, flags=re.S)
assert n == 1, "markers not found in DESIGN.md"
(ROOT / "DESIGN.md").write_text(text)
print("\n".join(table) + tail)
