"""friture_amd/data/octave_filters.npz holds the reference's design numbers verbatim
(tools/extract_reference_tables.py); its digest was recorded next to the reference
(tests/golden/filter_tables.sha256), and our own re-derivation stays close."""
import hashlib
from pathlib import Path

import numpy as np

from friture_amd import filter_design

ROOT = Path(__file__).resolve().parents[1]


def test_table_digest():
    tabs = filter_design.load_tables()
    want = dict(reversed(l.split("  ")) for l in (ROOT / "tests/golden/filter_tables.sha256").read_text().splitlines())
    assert sorted(want) == sorted(tabs)
    for k, v in tabs.items():
        assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() == want[k], k


def test_shapes_and_sizes():
    tabs = filter_design.load_tables()
    assert list(tabs["fft_sizes"]) == [1536, 1024, 768, 640, 576, 576, 540, 540, 540] == filter_design.ola_fft_sizes()
    assert tabs["bdec"].shape == tabs["adec"].shape == (13,)
    for bpo in filter_design.BANDS:
        assert tabs[f"boct_{bpo}"].shape == tabs[f"aoct_{bpo}"].shape == (bpo, 5)
        assert tabs[f"boct_fir_{bpo}"].shape == (bpo, 512)


def test_rederivation_is_close():
    """scipy's elliptic design drifted between versions: agreement is 1e-5..4e-3, not bit-exact —
    which is why the shipped table holds the reference's numbers, not the re-derived ones."""
    tabs = filter_design.load_tables()
    mine = filter_design.design_all()

    def rel(a, b):
        return np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(mine["bdec"], tabs["bdec"]) < 1e-3 and rel(mine["adec"], tabs["adec"]) < 1e-3
    for bpo in filter_design.BANDS:
        assert rel(mine[f"boct_{bpo}"], tabs[f"boct_{bpo}"]) < 1e-2
        assert rel(mine[f"aoct_{bpo}"], tabs[f"aoct_{bpo}"]) < 1e-3
        assert rel(mine[f"boct_fir_{bpo}"], tabs[f"boct_fir_{bpo}"]) < 1e-2
    # the minimum-phase construction itself is exact: feeding it the reference's IIR numbers
    # reproduces the reference's FIR taps
    fir = filter_design.minimum_phase_fir(tabs["bdec"], tabs["adec"])
    assert rel(fir, tabs["bdec_fir"]) < 1e-10
