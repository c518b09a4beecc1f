"""tools/bench_gcc.py on a variant library: FRT_LIB_VARIANT=<name> python tools/exp/gcc_variant_bench.py --pairs ..."""
import os, sys, runpy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from friture_amd import _lib
if os.environ.get("FRT_LIB_VARIANT"):
    _lib.LIB_PATH = Path(__file__).resolve().parents[1] / "variants" / os.environ["FRT_LIB_VARIANT"] / "libfriture_hip.so"
sys.argv = [str(Path(__file__).resolve().parents[1] / "bench_gcc.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
