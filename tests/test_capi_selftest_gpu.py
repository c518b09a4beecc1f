"""The standalone C++ driver of the C ABI (tools/stft_selftest.cpp, built by friture_amd.build.build_tools): every FFT
size, hop class, precision, staging mode and output kind of frt_stft_run against a double-precision host FFT, without
Python in the loop.  It is part of the GPU gate so that a kernel change that breaks a corner only it covers (the float64
large-frame colour image did, once) cannot pass."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_cpp_selftest_of_the_stft_entry_points():
    exe = ROOT / "tools" / "bin" / "stft_selftest"
    if not exe.exists():
        from friture_amd import build
        build.build_tools(verbose=False)
    r = subprocess.run([str(exe), "check"], capture_output=True, text=True, timeout=600)
    failing = [line for line in r.stdout.splitlines() if line.startswith("FAIL")]
    assert r.returncode == 0 and not failing, "\n".join(failing[:20]) + "\n" + r.stdout[-500:] + r.stderr[-500:]
    assert "ok" in r.stdout
