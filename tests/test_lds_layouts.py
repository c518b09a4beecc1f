"""The LDS exchange layouts of round 4's kernels against the bank rules of gfx950 (MI355X_MICROARCH.md §LDS), on the CPU:
tools/exp/lds_layout_model.py restates every address formula of csrc/ola_wave.h and csrc/stft_pk16*.h and counts, per
instruction and lane group, the distinct addresses that meet on one bank.  What DESIGN.md claims — every exchange access of
these kernels is conflict free, the mirrored reads of the unpack at most two-way (one lane pair per group where bin 16 j and its
neighbours change slot) — is asserted here; the measured counterpart is SQ_LDS_BANK_CONFLICT = 0 for ola_pair_kernel
(profiles/r04_ola_pair_pmc.txt) and 7 % of SQ_LDS_IDX_ACTIVE for the large-frame family (its sample ring included)."""
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("lds_layout_model", ROOT / "tools" / "exp" / "lds_layout_model.py")
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)


def test_bank_model_itself():
    # 64 lanes reading consecutive 8-byte words: conflict free; all lanes of a group on one bank at different addresses: 32-way
    assert model.degree("read_b64", [8 * l for l in range(64)]) == 1
    assert model.degree("read_b64", [256 * l for l in range(64)]) == 32
    assert model.degree("write_b64", [128 * l for l in range(64)]) == 16
    # identical addresses broadcast
    assert model.degree("read_b64", [64] * 64) == 1
    # 16-byte accesses: lanes 0-7 of a store group on the same four banks
    assert model.degree("write_b128", [128 * l for l in range(64)]) == 8
    assert model.degree("read_b128", [16 * l for l in range(64)]) == 1


@pytest.mark.parametrize("kernel", sorted(model.KERNELS))
def test_exchange_layouts_are_conflict_free(kernel):
    res = model.KERNELS[kernel]()
    for access, deg in res.items():
        if access == "unpack read (M - k)":
            assert deg <= 2, (kernel, access, deg)
        else:
            assert deg == 1, (kernel, access, deg)


def test_source_and_model_agree_on_the_constants():
    """the strides and offsets the model uses appear in the sources (a changed layout must change the model)"""
    src = {n: (ROOT / "friture_amd" / "csrc" / n).read_text() for n in ("stft_pk16.h", "stft_pk16s.h", "ola_wave.h")}
    assert "RS = MS + 34" in src["stft_pk16.h"] and "(lv + 272 * lu) * 8" in src["stft_pk16.h"] and "(17 * lv + 272 * lu) * 8" in src["stft_pk16.h"]
    # the size template (N = 8192 / 4096 / 2048): region stride and bases per lane count, the exchange maps
    s = src["stft_pk16s.h"]
    assert "RS = L == 16 ? MS + 34 : MS + 32" in s and "wave + 4 * (qw >> 1) + 8 * (qw & 1)" in s and "(17 * lp) * 8" in s
    assert "RS * reg + 8 * ((reg + (reg >> 2)) & 3)" in s and "(lp ^ c) * 8u" in s
    assert "RS * reg + 4 * ((reg & 5) | ((((reg >> 1) ^ (reg >> 3)) & 1) << 1))" in s
    assert "(t & 7) + 128 * (t >> 3)" in src["ola_wave.h"] and "(k2h ^ b) + 128 * k2l + 256 * b" in src["ola_wave.h"] and "(t ^ bb) + 128 * (e + 2 * bb)" in src["ola_wave.h"]
