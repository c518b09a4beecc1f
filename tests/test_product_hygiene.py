"""The product never routes through the oracle or any CPU fallback."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def product_sources():
    for p in (ROOT / "friture_amd").rglob("*"):
        if p.suffix in {".py", ".hip", ".h", ".cpp"} and "lib" not in p.relative_to(ROOT / "friture_amd").parts[:1]:
            yield p


def test_product_does_not_import_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]dsp|refshim|/root/reference", re.M)
    offenders = [str(p) for p in product_sources() if pat.search(p.read_text())]
    assert not offenders, offenders


def test_no_compat_layers_in_kernels():
    pat = re.compile(r"__HIP_PLATFORM_AMD__|__CUDACC__|cuda_runtime|hipify|triton", re.I)
    offenders = [str(p) for p in product_sources() if p.suffix in {".hip", ".h", ".cpp"} and pat.search(p.read_text())]
    assert not offenders, offenders


def test_runtime_entry_points_do_not_read_the_reference_checkout():
    for name in ("bench.py", "__graft_entry__.py"):
        p = ROOT / name
        if p.exists():
            assert "/root/reference" not in p.read_text().replace("when `/root/reference`", "")


def test_the_library_never_reads_the_environment():
    """No getenv in the shipped library: the only one sits inside the -DFRT_EXPERIMENTS helper of common.h (tools/exp builds); path
    selection for tests goes through frt_set_option.  The built library must not import getenv at all."""
    import subprocess
    hits = []
    for p in product_sources():
        if p.suffix in {".hip", ".h", ".cpp"}:
            for n, line in enumerate(p.read_text().splitlines(), 1):
                if re.search(r"\bgetenv\s*\(", line) and not line.lstrip().startswith("//"):
                    hits.append(f"{p.name}:{n}")
    assert len(hits) == 1 and hits[0].startswith("common.h"), hits
    from friture_amd import _lib
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    assert not re.search(r"\b(secure_)?getenv\b", syms), "libfriture_hip.so imports getenv"


def test_library_size_and_experiment_kernels_stay_out():
    """The shipped library carries no superseded or experimental kernel generation (those compile only with -DFRT_EXPERIMENTS:
    stft_pk_kernel, stft_pk16r_kernel, ola_batch_kernel) and stays below 2.5 MB (device code compressed)."""
    import subprocess
    from friture_amd import _lib
    assert _lib.LIB_PATH.stat().st_size < 2_500_000, _lib.LIB_PATH.stat().st_size
    syms = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    for name in ("stft_pk_kernel", "stft_pk16r_kernel", "ola_batch_kernel"):
        assert name not in syms, name
