#!/bin/bash
timeout 900 python -m pytest tests/test_gcc_gpu.py tests/test_sharding_gpu.py -x -q 2>&1 | tail -3
python - <<'PY'
import time, numpy as np, torch, os, sys
sys.path.insert(0, os.getcwd())
from friture_amd import _lib
from friture_amd.signal.correlation import GccPhat
_lib.init(0)
for pairs in (1, 16, 100, 128, 256, 1024):
    rng = np.random.default_rng(1)
    d0 = torch.from_numpy(0.25 * rng.standard_normal((pairs, 24000))).cuda()
    d1 = torch.roll(d0, 37, dims=1).contiguous()
    for mode in ("auto", "one"):
        if mode == "one": os.environ["FRT_GCC_ONE_WORKGROUP"] = "1"
        else: os.environ.pop("FRT_GCC_ONE_WORKGROUP", None)
        g = GccPhat(24000, pairs)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            g.correlate(d0, d1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.correlate(d0, d1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"pairs={pairs} {mode}: {dt*1e3:.3f} ms  {pairs/dt:.3e} windows/s")
PY
