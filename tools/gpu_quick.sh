#!/bin/bash
R=$GRAFT_REPO_ROOT
cat > /tmp/pb.py <<PY
import sys, time; sys.path.insert(0,"$R")
import numpy as np, torch
from friture_amd.pitch_tracker import PitchEngine
x=torch.from_numpy(0.2*np.random.default_rng(0).standard_normal((8,1<<22))).cuda()
for lim in (1<<30, 470<<20, 320<<20, 240<<20, 160<<20):
    e=PitchEngine(4096,1024,8); e.set_scratch_limit(lim)
    for _ in range(2): e.track(x)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): e.track(x)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(lim>>20, "MB: %.3f ms"%(dt*1e3), "%.3e frames/s"%(8*e.frames_for(1<<22)/dt))
PY
python /tmp/pb.py
