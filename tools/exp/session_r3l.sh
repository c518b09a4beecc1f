#!/bin/bash
# round 3, session l: mixed-radix FFT with per-pass twiddle tables (GCC-PHAT, streaming FIR bank)
timeout 600 python -m pytest tests/test_gcc_gpu.py tests/test_ola_gpu.py tests/test_widgets_gpu.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from friture_amd import _lib
from friture_amd.signal.correlation import GccPhat
_lib.init(0)
for pairs in (1, 100, 1024):
    L = 24000
    rng = np.random.default_rng(1)
    d0 = torch.from_numpy(0.25 * rng.standard_normal((pairs, L))).cuda()
    d1 = torch.roll(d0, 37, 1).contiguous()
    g = GccPhat(L, pairs)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        g.correlate(d0, d1); torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): g.correlate(d0, d1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"gcc pairs {pairs}: {dt*1e3:.4f} ms  {pairs/dt:.4g} windows/s")
PY
