"""The C-ABI library loads without a GPU and exports exactly what include/friture_hip.h declares."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "friture_hip.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(frt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from friture_amd import _lib
    return _lib.load()


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "frt_init" in syms and "frt_stft_run" in syms and len(syms) >= 14


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in friture_hip.h but not exported"


def test_binding_table_matches_header(lib):
    from friture_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_exports_are_c_abi_only():
    from friture_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "frt_" in l.split()[-1][:4])
    assert exported == declared_symbols()


def test_no_gpu_calls_fail_loudly_not_silently(lib):
    """Without a device frt_init reports an error string; with one it succeeds.  Never a fallback."""
    from friture_amd import _lib
    rc = lib.frt_init(0, None, None)
    if rc != 0:
        assert rc == -3 and b"frt_init" in lib.frt_last_error()
        with pytest.raises(_lib.FritureHipError):
            _lib.check(rc)


def test_path_selection_options_need_no_gpu_and_reject_unknown_names(lib):
    """frt_set_option / frt_get_option (the library never reads the environment): the documented names round-trip, a
    negative value restores the shape rule (-1), anything else is FRT_ERR_INVALID with a message."""
    from friture_amd import _lib
    for name in ("gcc_one_workgroup", "gcc_any_length", "ola_chunk_kernels", "pitch_grid_two_pass", "gcc_resident", "iir_lookback", "ola_defer", "iir_lane_columns"):
        assert _lib.get_option(name) == -1
        _lib.set_option(name, 1)
        assert _lib.get_option(name) == 1
        _lib.set_option(name, -7)
        assert _lib.get_option(name) == -1
    with pytest.raises(_lib.FritureHipError) as err:
        _lib.set_option("FRT_EDGE_SCALE", 0)
    assert err.value.status == -1 and "unknown option" in str(err.value)
