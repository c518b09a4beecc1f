"""Is a kernel clocked by the chip's power budget?  The same launches on the bench's noise input, on a tone and on an all-zero
input (identical instruction streams and memory traffic; the data toggles fewer bits): python tools/exp/clock_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from friture_amd import _lib, tables, palette
from friture_amd.stft import StftEngine
_lib.init(0)
def run(n_fft, hop, C, log2t, kind, fill):
    T = 1 << log2t
    xs = []
    for b in range(3):
        if fill == "noise": x = 0.25 * torch.randn((C, T), device="cuda", dtype=torch.float32)
        elif fill == "tone": x = (0.5 * torch.sin(2 * np.pi * 1000.0 / 48000.0 * torch.arange(T, device="cuda", dtype=torch.float32))).repeat(C, 1).contiguous()
        else: x = torch.zeros((C, T), device="cuda", dtype=torch.float32)
        xs.append(x)
    e = StftEngine(n_fft, hop, C, 32)
    e.set_epilogue(tables.weighting_db(tables.rfft_frequencies(n_fft), 1e-50)[0], -140.0, 0.0, palette.cmr_lut())
    F = e.frames_for(T)
    outs = [torch.empty((C, F, n_fft // 2 + 1), dtype=torch.int32 if kind == 3 else torch.float32, device="cuda") for _ in range(3)]
    import time
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 0.3:
        e.run(kind, xs[k % 3], outs[k % 3]); k += 1
        if k % 8 == 0: torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for k in range(60): e.run(kind, xs[k % 3], outs[k % 3])
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / 60
for cfg in ((1024, 512, 1, 26, 3), (1024, 512, 1, 26, 0), (16384, 8192, 32, 20, 0), (16384, 8192, 32, 20, 3)):
    r = {f: run(*cfg, f) for f in ("noise", "tone", "zeros")}
    print(f"N {cfg[0]} hop {cfg[1]} C {cfg[2]} kind {cfg[4]}: ms per launch  noise {r['noise']:.4f}  tone {r['tone']:.4f}  zeros {r['zeros']:.4f}")
