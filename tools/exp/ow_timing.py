"""Cycle counts between the phases of one wavefront's second filter in ola_wave_kernel, from a -DFRT_OW_TIMING=1 variant library:
FRT_LIB_VARIANT=<name> python tools/exp/ow_timing.py [channels bpo log2n]"""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from friture_amd import _lib
if os.environ.get("FRT_LIB_VARIANT"):
    _lib.LIB_PATH = Path(__file__).resolve().parents[1] / "variants" / os.environ["FRT_LIB_VARIANT"] / "libfriture_hip.so"
from friture_amd import filter_design
from friture_amd.filter import FirBank
ch, bpo, log2n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 3, 22)
_lib.init(0)
t = filter_design.load_tables()
n = 1 << log2n
x = 0.25 * torch.randn((ch, n), device="cuda", dtype=torch.float32)
decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
out = torch.empty((ch, n // 1024, 9 * bpo), dtype=torch.float32, device="cuda")
bank = FirBank(bpo, ch, t)
lib = ctypes.CDLL(str(_lib.LIB_PATH))
names = ["spectrum x H (table loads, register moves)", "radix 32 + twiddles", "exchange 1", "radix 8 + twiddles", "exchange 2", "radix 8",
         "outputs + energies"]
acc = []
for _ in range(6):
    bank.energies(x, 1024, alphas, out=out)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.frt_ow_timing_read(buf) == 0
    v = np.array(buf[:8], dtype=np.int64)
    acc.append(np.diff(v))
acc = np.array(acc[1:])
print("cycles (s_memtime ticks), median of 5 calls")
for i, nme in enumerate(names):
    print(f"{nme:46s} {int(np.median(acc[:, i])):8d}")
print(f"{'one filter':46s} {int(np.median(acc.sum(axis=1))):8d}")
