#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/g.py <<PY
import sys, time; sys.path.insert(0,".")
import numpy as np, torch
from friture_amd.signal.correlation import GccPhat
from oracle import dsp
rng=np.random.default_rng(1)
for pairs in (100, 1024, 4096):
    d0=0.25*rng.standard_normal((pairs,24000)); d1=np.roll(d0,37,axis=1)+0.025*rng.standard_normal((pairs,24000))
    a0,a1=torch.from_numpy(d0).cuda(),torch.from_numpy(d1).cuda()
    g=GccPhat(24000,pairs)
    g.correlate(a0,a1); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): x,am=g.correlate(a0,a1)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    ref,_,_=dsp.gcc_phat(d0[0].copy(),d1[0].copy())
    err=np.max(np.abs(x[0].cpu().numpy()-ref))/np.max(np.abs(ref))
    print(pairs, "pairs: %.3f ms  %.3e windows/s  err %.2e argmax %d"%(dt*1e3, pairs/dt, err, int(am[0])))
PY
echo "auto"; python /tmp/g.py
echo "split forced off"; FRT_GCC_SPLIT=0 python /tmp/g.py
python -m pytest tests/test_gcc_gpu.py -x -q 2>&1 | tail -2
FRT_GCC_SPLIT=1 python -m pytest tests/test_gcc_gpu.py -x -q 2>&1 | tail -2
