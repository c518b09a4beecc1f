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


HOSTILE_ENV = {"FRT_EDGE_SCALE": "0", "FRT_ABLATE": "15", "FRT_STFT_NO_DMA": "1", "FRT_STFT_NO_RING": "1", "FRT_STFT_RING_IMAGE": "1",
               "FRT_STFT_NO_PK": "1", "FRT_STFT_NO_PK16": "1", "FRT_STFT_NO_PK16H": "1", "FRT_STFT_NO_PK16Q": "1", "FRT_STFT_NO_PK16W": "1",
               "FRT_GCC_FORCE_R": "4", "FRT_GCC_FORCE_ANY": "1", "FRT_GCC_ONE_WORKGROUP": "0", "FRT_GCC_NO_STATIC_PLAN": "1", "FRT_GCC_PROFILE": "1",
               "FRT_IIR_NO_LANE_KERNEL": "1", "FRT_IIR_EXACT_OPS": "1", "FRT_IIR_LONG_SCAN_ROWS": "1", "FRT_ZS_VECTOR": "1",
               "FRT_ZS_MAX_SLICES": "1", "FRT_ZS_WAVE_GOAL": "1", "FRT_NO_GRAPH": "1", "FRT_OLA_NO_WAVE": "1", "FRT_OLA_NO_CHUNK_KERNELS": "1",
               "FRT_PITCH_GRID_2PASS": "1"}

_DIGEST_SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from friture_amd import _lib, filter_design, palette, tables
from friture_amd.filter import FirBank, IirBank
from friture_amd.signal.correlation import GccPhat
from friture_amd.stft import StftEngine
_lib.init(0)
rng = np.random.default_rng(99)
h = hashlib.sha256()
for n_fft, hop, C, T in ((1024, 512, 2, 1 << 16), (16384, 4096, 2, 1 << 17), (4096, 2048, 1, 1 << 16)):
    x = (0.25 * rng.standard_normal((C, T))).astype(np.float32)
    e = StftEngine(n_fft, hop, C, 32)
    e.set_epilogue(tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0], -140.0, 0.0, palette.cmr_lut())
    h.update(e.image(x).tobytes()); h.update(e.psd(x).tobytes())
t = filter_design.load_tables()
xb = (0.25 * rng.standard_normal((2, 1 << 15))).astype(np.float32)
al = np.full(27, 0.01)
ib = IirBank(t["bdec"], t["adec"], list(t["boct_3"]), list(t["aoct_3"]), 2); ib.set_chunk(1024)
h.update(ib.energies(xb, 1024, al).tobytes())
h.update(FirBank(3, 2, t).energies(xb, 1024, al).tobytes())
d0 = 0.25 * rng.standard_normal((3, 24000)); d1 = np.roll(d0, 5, axis=1)
xc, am = GccPhat(24000, 3).correlate(d0, d1)
h.update(np.asarray(xc).tobytes()); h.update(np.asarray(am).tobytes())
print(h.hexdigest())
'''


@pytest.mark.gpu
def test_outputs_do_not_depend_on_the_environment():
    """One image + PSD per kernel family of K1, one call of each bank and one GCC-PHAT batch, in a fresh process with every
    experiment knob the library ever had set to a hostile value, against a fresh process with a clean environment: the same
    bytes (round 4's library let FRT_EDGE_SCALE rescale the colour-index threshold of every frt_stft_run)."""
    import os
    import subprocess
    import sys
    clean = {k: v for k, v in os.environ.items() if not k.startswith("FRT_")}
    a = subprocess.run([sys.executable, "-c", _DIGEST_SCRIPT], cwd=str(ROOT), env=clean, capture_output=True, text=True)
    b = subprocess.run([sys.executable, "-c", _DIGEST_SCRIPT], cwd=str(ROOT), env={**clean, **HOSTILE_ENV}, capture_output=True, text=True)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-1500:], b.stderr[-1500:])
    assert a.stdout.strip().splitlines()[-1] == b.stdout.strip().splitlines()[-1]
