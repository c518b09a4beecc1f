"""The ring instance of the headline STFT kernel (stft.hip, SHIFT = -1) decides that an LDS-DMA copy has landed by counting
the vector-memory operations younger than it: `s_waitcnt vmcnt(9)` is right only while every frame issues exactly nine row
stores behind the copy, for every output kind the instance serves.  This test reads the generated code and fails when a
compiler or epilogue change breaks that count (the alternative, a silent read of stale LDS, would show only as wrong
spectra on one toolchain)."""
import os
import re
import shutil
import subprocess
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def device_assembly():
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    src = ROOT / "friture_amd" / "csrc" / "stft.hip"
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "stft.s"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", f"-I{ROOT / 'include'}", f"-I{src.parent}",
                            "-S", "--cuda-device-only", "-o", str(out), str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out.read_text()


@pytest.mark.parametrize("split,stores", [(0, 9), (1, 8)])
def test_ring_instance_row_stores_per_frame_match_its_vmcnt(device_assembly, split, stores):
    """packed rows: nine stores per frame and vmcnt(9); split rows (frt_stft_run_split): eight — bin N/4 rides in lane 0's
    last descending store, the Nyquist bin waits in a register for the end of the run — and vmcnt(8).  (Round 5: the split
    instance first shipped with vmcnt(9) and read half-landed copies at full size; the GPU test caught it, this one pins it.)"""
    text = device_assembly
    m = re.search(rf"^(_ZN3frt11stft_kernelIffLi9ELin1ELb{split}EEEvNS_8StftArgsE):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M)
    assert m, "ring instance (float32, N = 1024, SHIFT = -1) not found"
    lines = [ln.strip() for ln in m.group(2).splitlines()]
    code = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) or ln.startswith(".LBB")]
    waits = [ln for ln in code if ln.startswith("s_waitcnt") and f"vmcnt({stores})" in ln]
    assert waits, f"the ring instance no longer waits with vmcnt({stores}): update this test together with the kernel"
    copies = [i for i, ln in enumerate(code) if ln.startswith("global_load_lds_dwordx4")]
    assert copies, "no LDS-DMA copy in the ring instance"
    # row stores come in clusters (one per output kind and frame phase): eight unconditional and the Nyquist bin's
    # (a cluster ends where a label stands between two stores: the guarded store of lane 0 is followed, not preceded, by its label)
    clusters, run, last = [], 0, None
    for i, ln in enumerate(code):
        if ln.startswith("global_store_dword "):
            if run and any(c.startswith(".LBB") for c in code[last + 1:i]):
                clusters.append(run)
                run = 0
            run += 1
            last = i
    if run:
        clusters.append(run)
    if split:                           # the run's Nyquist values: one store behind the frame loop
        assert clusters[-1] == 1, clusters
        clusters = clusters[:-1]
    assert clusters and all(c == stores for c in clusters), f"row stores per frame: {clusters} (vmcnt({stores}) assumes {stores})"
    # nothing else touches vector memory inside the frame loop
    first_loop_wait = next(i for i, ln in enumerate(code) if ln.startswith("s_waitcnt") and f"vmcnt({stores})" in ln)
    other = [ln for ln in code[first_loop_wait:] if re.match(r"(global|buffer|flat)_(load|atomic)", ln) and "lds" not in ln]
    assert not other, f"other vector-memory operations inside the frame loop: {other[:4]}"
