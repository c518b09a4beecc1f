"""The committed evidence must belong to the committed code (the driver's line carried
`roofline.traffic: null` because stft.hip was edited after the PMC session that wrote profiles/pmc_traffic.json)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def test_pmc_traffic_was_measured_on_these_kernel_sources():
    import bench
    rec = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    assert rec["kernel_sources"] == bench.kernel_source_digest(), (
        "profiles/pmc_traffic.json was measured on other kernel sources: re-run the PMC passes of tools/gpu_session.sh "
        "(tools/pmc_traffic.py) on the GPU and commit the refreshed file, or bench.py prints roofline.traffic = null")
    assert (rec["n_fft"], rec["hop"], rec["frames"]) == (1024, 512, 131071)
    # sanity of the figure itself: at least the algorithmic bytes, at most 1.25x (wasted re-reads would show here)
    assert 1.0 <= rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"] <= 1.25


def test_leg_traffic_was_measured_on_these_kernel_sources():
    """profiles/r06_leg_traffic.json (tools/leg_traffic.py: rocprofv3 PMC passes of every bench leg's workload) carries, per leg, the
    digest of the kernel sources it was measured on: bench.py quotes a leg's roofline.traffic only while that digest matches, so a
    kernel edit without a fresh PMC session fails here instead of silently printing traffic = null."""
    import bench
    rec = json.loads((ROOT / "profiles" / "r06_leg_traffic.json").read_text())["legs"]
    served = {name for r in rec.values() for name in r["bench_legs"]}
    assert served == set(bench.LEG_SOURCES), sorted(set(bench.LEG_SOURCES) - served)
    for leg, r in rec.items():
        for name in r["bench_legs"]:
            assert r["kernel_sources"] == bench.sources_digest(bench.LEG_SOURCES[name]), (
                f"{leg}: measured on other kernel sources — re-run tools/gpu_leg_traffic.sh on the GPU and commit the refreshed file")
            assert bench.leg_traffic(name) == r["hbm_bytes_per_call"] > 0
    # the default-window GCC-PHAT kernel keeps everything between the signals and the correlation on the CU: traffic within 15 % of
    # the algorithmic bytes
    g = rec["gcc1024"]
    assert 1.0 <= g["hbm_bytes_per_call"] / g["algorithmic_bytes_per_call"] <= 1.15
