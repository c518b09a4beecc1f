"""bench.py prints ONE JSON line and the driver keeps about the last 8 KB of stdout: the printed line of a full run (every leg,
parity, cpu_baseline, ranks_seen) must stay under bench.LINE_BUDGET characters.  Checked on the full record of this round's GPU run
(profiles/r06_bench_full.json, written by --full-json) pushed through the same compaction the line goes through."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def test_compacted_line_of_a_full_run_fits_the_drivers_tail():
    import bench
    full = json.loads((ROOT / "profiles" / "r06_bench_full.json").read_text())
    assert len(full["legs"]) >= 13 and "parity" in full and "cpu_baseline" in full
    compact = bench.compact_line(full)
    line = json.dumps(compact, separators=(",", ":"))
    assert len(line) < bench.LINE_BUDGET, len(line)
    # nothing the contract or the review reads is lost
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "parity", "ranks_seen", "legs", "octave_bands"):
        assert key in compact, key
    assert compact["value"] == full["value"] and compact["ms_per_step"] == full["ms_per_step"]
    for name, leg in full["legs"].items():
        c = compact["legs"][name]
        assert abs(c["value"] / leg["value"] - 1) < 1e-4
        if "roofline" in leg:
            assert abs(c["roofline"]["frac"] / leg["roofline"]["frac"] - 1) < 1e-4
        if "parity" in leg:
            assert c["parity"]["gate"] == leg["parity"]["gate"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "host_fixed_us"):
        assert k in compact["roofline"]
    assert compact["layout"] == "split" and "packed_rows" in compact
    # every leg with a kernel of its own carries measured traffic (profiles/r06_leg_traffic.json)
    for name in bench.LEG_SOURCES:
        assert compact["legs"][name]["roofline"]["traffic"] > 0, name
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in compact["cpu_baseline"]


def test_prose_keys_leave_the_line_but_stay_in_the_full_record():
    import bench
    rec = {"value": 1.23456789, "ms_per_step": 0.123456789, "legs": {"a": {"value": 2.0, "mode": "prose", "roofline": {"frac": 0.123456, "model": "prose"}}},
           "roofline": {"frac": 0.5, "traffic_note": "prose"}, "note": "prose"}
    c = bench.compact_line(rec)
    assert "note" not in c and "mode" not in c["legs"]["a"] and "model" not in c["legs"]["a"]["roofline"] and "traffic_note" not in c["roofline"]
    assert c["value"] == rec["value"] and c["legs"]["a"]["roofline"]["frac"] == 0.12346
    assert "note" in rec and "mode" in rec["legs"]["a"]                       # the input is not modified
