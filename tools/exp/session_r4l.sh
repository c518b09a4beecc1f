#!/bin/bash
# round 4, session l: first stage of frame g + 1 inside frame g's sub-transform phase (PSD kind)
set -u
mkdir -p gpurun_out
echo "== parity"; timeout 300 python tools/exp/pk_debug.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q -m gpu -k "large_frame or lds_staged or all_sizes or randomised" -p no:cacheprovider 2>&1 | tail -3
FRT_LIB_VARIANT=pktime timeout 200 python tools/exp/pk_timing.py 8192 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4l_timing.txt
echo "== variants"
bash tools/exp/ab_variants.sh "base noovl base noovl" "16384 8192 32 20 0 0 40" "16384 4096 32 20 0 0 40" 2>&1 | tee gpurun_out/r4l_ab.txt
