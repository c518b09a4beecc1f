"""The output passes of the exact IIR bank (iir.hip: iir_lane_kernel, iir_lane_wg4_kernel, iir_lane_split_kernel) request a lane's next
sixteen samples one trip ahead.  From round 3 to round 6 the compiler waited for those requests right where it issued them (the
request stood behind an `if`; the copies of the register merge landed in the conditional block: global_load x4, s_waitcnt vmcnt(3..0),
v_mov x16 at the top of every trip) and nobody noticed: the kernels were correct, and longer prefetch distances simply "measured equal".
This test reads the generated code: in the main loops of those kernels every global sample load must have at least one multiply-add
between itself and the wait that covers it.  The column form (iir_lane_col_kernel) copies its samples by LDS-DMA from inline assembly and
waits with vmcnt(0) by hand in front of the tile barrier: checked for the copy, the wait and the barrier in that order inside its loops."""
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
    from friture_amd import build
    src = ROOT / "friture_amd" / "csrc" / "iir.hip"
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "iir.s"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *build.EXTRA_FLAGS["iir.hip"], f"-I{ROOT / 'include'}", f"-I{src.parent}",
                            "-S", "--cuda-device-only", "-o", str(out), str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out.read_text()


def kernel_code(text, mangled_prefix):
    out = {}
    for m in re.finditer(rf"^({mangled_prefix}\w*):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        lines = [ln.strip() for ln in m.group(2).splitlines()]
        out[m.group(1)] = [ln for ln in lines if ln and not ln.startswith((";", "//")) and (not ln.startswith(".") or ln.startswith(".LBB"))]
    return out


def inner_loops(code):
    labels = {ln.split(":")[0]: i for i, ln in enumerate(code) if ln.startswith(".LBB")}
    loops = []
    for i, ln in enumerate(code):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    return [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]


def is_vm(ln):
    return ln.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "scratch_", "flat_load", "flat_store"))


@pytest.mark.parametrize("prefix", ["_ZN3frt15iir_lane_kernel", "_ZN3frt19iir_lane_wg4_kernel", "_ZN3frt21iir_lane_split_kernel"])
def test_sample_requests_of_the_output_passes_run_ahead_of_their_wait(device_assembly, prefix):
    kernels = kernel_code(device_assembly, prefix)
    assert len(kernels) == 2, sorted(kernels)                      # float32 and float64 input
    checked = 0
    for name, code in kernels.items():
        assert not any(ln.startswith("scratch_") for ln in code), f"{name}: scratch traffic"
        for a, b in inner_loops(code):
            body = [ln for ln in code[a:b + 1] if not ln.startswith(".LBB")]
            fma = sum(1 for ln in body if ln.startswith(("v_fma_f64", "v_fmac_f64")))
            loads = [i for i, ln in enumerate(body) if ln.startswith("global_load_dwordx4")]
            if fma < 150 or not loads:                             # the sample loops: 16 samples x (2 ORD + 1) multiply-adds per trip and filter
                continue
            vm = [i for i, ln in enumerate(body) if is_vm(ln)]
            for i in loads:
                issued = sum(1 for j in vm if j <= i)
                covered_at = None
                for j in range(i + 1, len(body)):
                    if is_vm(body[j]):
                        issued += 1
                    m = re.search(r"vmcnt\((\d+)\)", body[j]) if body[j].startswith("s_waitcnt") else None
                    if m and issued - int(m.group(1)) >= sum(1 for q in vm if q <= i):
                        covered_at = j
                        break
                if covered_at is None:                             # covered at the top of the next trip: a whole trip of arithmetic in between
                    covered_at = len(body) + next((j for j, ln in enumerate(body) if ln.startswith("s_waitcnt") and "vmcnt" in ln), 0)
                between = [ln for ln in (body + body)[i + 1:covered_at]]
                work = sum(1 for ln in between if ln.startswith(("v_fma_f64", "v_fmac_f64")))
                assert work >= 100, f"{name}: a sample request is awaited {covered_at - i - 1} instructions ({work} multiply-adds) behind its issue"
                checked += 1
    assert checked >= 8, checked                                   # 4 or 8 loads per trip in the band and the decimator loop of both instances


def test_column_form_waits_for_its_lds_copies_in_front_of_the_tile_barrier(device_assembly):
    kernels = kernel_code(device_assembly, "_ZN3frt19iir_lane_col_kernel")
    assert len(kernels) == 2, sorted(kernels)
    for name, code in kernels.items():
        assert not any(ln.startswith(("scratch_", "flat_")) for ln in code), f"{name}: scratch or flat accesses (the tiles must be LDS accesses)"
        copies = [i for i, ln in enumerate(code) if ln.startswith("global_load_lds_dwordx4")]
        assert len(copies) >= 4, (name, len(copies))
        barriers = [i for i, ln in enumerate(code) if ln.startswith("s_barrier")]
        assert barriers
        for b in barriers:                                         # every tile barrier: a vmcnt(0) directly in front of it
            front = [ln for ln in code[max(0, b - 6):b] if ln.startswith("s_waitcnt")]
            assert any("vmcnt(0)" in ln for ln in front), (name, code[max(0, b - 6):b + 1])
        assert sum(1 for ln in code if ln.startswith("ds_read_b128")) >= 12, name
