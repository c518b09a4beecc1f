"""Build recipe for libfriture_hip.so (hand-written HIP for gfx950) and the C oracle.

`python -m friture_amd.build` compiles every translation unit under friture_amd/csrc with hipcc
(cross-compiles without a GPU) and links them in-tree as friture_amd/lib/libfriture_hip.so, so the
built library travels with the source snapshot to the GPU box.  `__graft_entry__.build()` calls
`build_all()`.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = PKG / "lib" / "obj"
LIB = LIBDIR / "libfriture_hip.so"
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXXFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "--offload-compress",
            f"-I{ROOT / 'include'}"]

# translation units with special flags: the exact IIR bank replays the reference's IEEE operation
# order (no fused multiply-add contraction) so that it can be bit-identical to lfilter.py:131-139
# stft.hip: packed fp32 VALU instructions (v_pk_fma_f32 ...) issue slower than the scalar pair they
# replace on gfx950 (measured: +7 % kernel time), so SLP packing of the butterflies is off
# iir.hip, -amdgpu-mfma-vgpr-form: the float64 MFMA accumulators of iir_zero_state_mfma_kernel in vector registers — without it the
# compiler keeps them in accumulation registers inside the loop and in vector registers across its back-edge: 16 v_accvgpr_write at
# the top and 16 v_accvgpr_read (+ s_nop 8) at the bottom of every trip of 16 MFMAs (profiles/r06_iir_zero_state_pipeline.txt: -3 .. -6 us per call)
EXTRA_FLAGS = {"iir.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form"], "pipeline.hip": ["-ffp-contract=off"], "pitch.hip": ["-ffp-contract=off"],
               "specgram.hip": ["-ffp-contract=off"],
               "stft.hip": ["-fno-slp-vectorize", "-Wno-inline-asm"]}


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _headers() -> list[Path]:
    return sorted(CSRC.glob("*.h")) + sorted((ROOT / "include").glob("*.h"))


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _compile(src: Path) -> Path:
    obj = OBJDIR / (src.stem + ".o")
    if _stale(obj, [src, Path(__file__)] + _headers()):
        cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_lib(verbose: bool = True) -> Path:
    if not Path(HIPCC).exists():
        raise RuntimeError(f"hipcc not found ({HIPCC}); cannot build libfriture_hip.so")
    OBJDIR.mkdir(parents=True, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs),
               "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[friture_amd.build] linked {LIB} from {len(objs)} objects", file=sys.stderr)
    return LIB


def build_oracle(verbose: bool = True) -> None:
    """Compile oracle/'s C restatement (test infrastructure only, never loaded by the product)."""
    mk = ROOT / "oracle" / "Makefile"
    if mk.exists():
        r = subprocess.run(["make", "-s", "-C", str(ROOT / "oracle")], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{r.stdout}\n{r.stderr}")


def build_tools(verbose: bool = True) -> None:
    """Standalone C++ drivers under tools/ that exercise the C ABI without Python."""
    tools = ROOT / "tools"
    (tools / "bin").mkdir(exist_ok=True)
    for src in sorted(tools.glob("*.cpp")):
        exe = tools / "bin" / src.stem
        if _stale(exe, [src, LIB] + _headers()):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O2", "-std=c++17", f"-I{ROOT / 'include'}", str(src),
                   "-o", str(exe), f"-L{LIBDIR}", "-lfriture_hip", "-Wl,-rpath,$ORIGIN/../../friture_amd/lib",
                   "-Wl,-rpath,/opt/rocm/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"tool build failed for {src.name}:\n{r.stdout}\n{r.stderr}")


def build_all(verbose: bool = True) -> Path:
    lib = build_lib(verbose)
    build_oracle(verbose)
    build_tools(verbose)
    return lib


if __name__ == "__main__":
    print(build_all())
