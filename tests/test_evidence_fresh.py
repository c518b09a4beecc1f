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
