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
