#!/bin/bash
# round 6, session e: parity after the advisor fixes and the fast PHAT weights; GCC rates
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== tests"; timeout 1500 python -m pytest tests/test_gcc_gpu.py tests/test_ola_gpu.py tests/test_iir_gpu.py tests/test_soak_gpu.py tests/test_stft_gpu.py -x -q 2>&1 | tail -4
echo "== gcc batch rates"
timeout 600 python tools/bench_gcc.py --pairs 1 64 100 256 1024 2>&1 | grep -v "^{" | grep -v amdgpu.ids | tee $O/gcc_batch.txt
