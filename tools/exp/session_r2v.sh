#!/bin/bash
python - <<'P'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from friture_amd import _lib
from friture_amd.pitch_tracker import PitchEngine, swipe_tables
_lib.init(0)
grid, _, kern = swipe_tables()
T = 1 << 22
tt = np.arange(T)
ph = 2 * np.pi * np.cumsum(110.0 * 2 ** (2.0 * tt / T)) / 48000.0
base = 0.2 * (np.sin(ph) + 0.6 * np.sin(2 * ph) + 0.3 * np.sin(3 * ph))
rng = np.random.default_rng(0)
x = torch.from_numpy(np.stack([base + 1e-3 * rng.standard_normal(T) for _ in range(8)])).cuda()
eng = PitchEngine(4096, 1024, 8, grid=grid, kernels=kern)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    eng.track(x); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): eng.track(x)
    torch.cuda.synchronize()
    print("pitch track ms:", (time.perf_counter() - t0) / 10 * 1e3)
P
