import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, time
from friture_amd import _lib, filter_design
from friture_amd.filter import IirBank
from pathlib import Path
if os.environ.get('FRT_LIB_VARIANT'):
    _lib.LIB_PATH = Path(__file__).resolve().parents[1] / 'variants' / os.environ['FRT_LIB_VARIANT'] / 'libfriture_hip.so'
_lib.init(0)
t = filter_design.load_tables()
for bpo, C, n in ((3, 8, 1 << 22), (24, 8, 1 << 20)):
    boct, aoct = list(t[f"boct_{bpo}"]), list(t[f"aoct_{bpo}"])
    decs = [2 ** j for j in range(9)[::-1] for _ in range(bpo)]
    alphas = np.array([1.0 - (1.0 - 0.65) ** (1.0 / (1.0 * 48000 / d + 1)) for d in decs])
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = 0.25 * torch.randn((C, n), generator=gen, device="cuda", dtype=torch.float32)
    ref = None
    for chunk in (2048, 1024, 512, 256):
        b = IirBank(t["bdec"], t["adec"], boct, aoct, C)
        b.set_chunk(chunk)
        out = torch.empty((C, n // 1024, 9 * bpo), dtype=torch.float32, device="cuda")
        e = b.energies(x, 1024, alphas, out=out).clone()
        e2 = b.energies(x, 1024, alphas, out=out).clone()      # a second call carries state
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            b.energies(x, 1024, alphas, out=out)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): b.energies(x, 1024, alphas, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        if ref is None: ref = (e, e2)
        d1 = float(((e - ref[0]).abs() / ref[0]).max()); d2 = float(((e2 - ref[1]).abs() / ref[1]).max())
        print(f"bpo {bpo} chunk {chunk}: {ms:.4f} ms  max rel diff to chunk 1024: first call {d1:.2e}, second call {d2:.2e}")
