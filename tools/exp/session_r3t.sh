#!/bin/bash
# phase times inside the one-workgroup GCC kernel (workgroup 0's stamps), 1 / 100 / 256 pairs
for p in 1 100 256; do
  FRT_GCC_PROFILE=1 FRT_GCC_ONE_WORKGROUP=1 timeout 120 python - $p <<'PY' 2>&1 | grep "phases" | tail -2
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from friture_amd.signal.correlation import GccPhat
pairs = int(sys.argv[1]); L = 24000
rng = np.random.default_rng(0)
d0 = 0.25 * rng.standard_normal((pairs, L)); d1 = np.roll(d0, 37, axis=1)
a0, a1 = torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()
g = GccPhat(L, pairs)
for _ in range(4): g.correlate(a0, a1)
torch.cuda.synchronize()
PY
done
