#!/bin/bash
# round 6, session l: the resident kernel without its last spilled register — parity, rates, traffic of the two GCC legs
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gcc_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_gcc.py --pairs 100 256 1024 2>&1 | grep -v "^{" | grep -v amdgpu.ids
bash tools/gpu_leg_traffic.sh r6l_traffic gcc1024 gcc100 2>&1 | tail -3
