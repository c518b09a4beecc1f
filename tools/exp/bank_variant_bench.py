"""tools/bench_firbank.py on a variant library: FRT_LIB_VARIANT=<name> python tools/exp/bank_variant_bench.py"""
import os, sys, runpy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from friture_amd import _lib
if os.environ.get("FRT_LIB_VARIANT"):
    _lib.LIB_PATH = Path(__file__).resolve().parents[1] / "variants" / os.environ["FRT_LIB_VARIANT"] / "libfriture_hip.so"
runpy.run_path(str(Path(__file__).resolve().parents[1] / "bench_firbank.py"), run_name="__main__")
