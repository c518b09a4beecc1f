"""A/B of the pitch tracker's log-grid kernels (register-resident vs two-pass) at sustained clocks; checks that both
give bit-identical tracks.   python tools/exp/pitch_ab.py"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from friture_amd import _lib
from friture_amd.pitch_tracker import PitchEngine, swipe_tables

_lib.init(0)
grid, _, kern = swipe_tables()
rng = np.random.default_rng(3)
sync = torch.cuda.synchronize
for n_fft, hop, ch, log2t in [(4096, 1024, 8, 22), (1024, 256, 8, 22)]:
    T = 1 << log2t
    tt = np.arange(T)
    ph = 2 * np.pi * np.cumsum(110.0 * 2 ** (2.0 * tt / T)) / 48000.0
    base = 0.2 * (np.sin(ph) + 0.6 * np.sin(2 * ph) + 0.3 * np.sin(3 * ph))
    x = torch.from_numpy(np.stack([base + 1e-3 * rng.standard_normal(T) for _ in range(ch)])).cuda()
    eng = PitchEngine(n_fft, hop, ch, grid=grid, kernels=kern)
    F = eng.frames_for(T)
    res = {}
    for rep in range(2):
        for name, env in (("reg", None), ("2pass", "1")):
            if env is None:
                os.environ.pop("FRT_PITCH_GRID_2PASS", None)
            else:
                os.environ["FRT_PITCH_GRID_2PASS"] = env
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                out = eng.track(x)
                sync()
            t0 = time.perf_counter()
            for _ in range(10):
                out = eng.track(x)
            sync()
            dt = (time.perf_counter() - t0) / 10
            res[name] = out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out)
            print(f"N={n_fft} hop={hop} {name}: {dt * 1e3:.3f} ms  {ch * F / dt:.3e} frames/s", flush=True)
    a, b = res["reg"], res["2pass"]
    print("bit-identical:", np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]))
    del x
