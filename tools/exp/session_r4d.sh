#!/bin/bash
# round 4, session d: staggered sub-transform rounds
set -u
mkdir -p gpurun_out
echo "== parity"; timeout 300 python tools/exp/pk_debug.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q -m gpu -k "large_frame or lds_staged" -p no:cacheprovider 2>&1 | tail -2
for a in "8192 0" "8192 3"; do FRT_LIB_VARIANT=pktime timeout 200 python tools/exp/pk_timing.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4d_timing.txt
echo "== variants"
bash tools/exp/ab_variants.sh "base nostag base nostag" "16384 8192 32 20 0 0 40" "16384 8192 32 20 3 0 40" "16384 4096 32 20 0 0 40" "16384 4096 32 20 3 0 40" 2>&1 | tee gpurun_out/r4d_ab.txt
