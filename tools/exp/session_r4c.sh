#!/bin/bash
# round 4, session c: pipelined unpack + priority alternation — parity, intervals, A/B
set -u
mkdir -p gpurun_out
echo "== parity"; timeout 300 python tools/exp/pk_debug.py 2>&1 | tail -2
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q -m gpu -k "large_frame or lds_staged or image_epilogue or db_norm" -p no:cacheprovider 2>&1 | tail -3
for v in pktime pktime2; do for a in "8192 0" "8192 3"; do echo "-- $v"; FRT_LIB_VARIANT=$v timeout 200 python tools/exp/pk_timing.py $a 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r4c_timing.txt
echo "== variants"
bash tools/exp/ab_variants.sh "base pkprio2 base pkprio2" "16384 8192 32 20 0 0 40" "16384 8192 32 20 3 0 40" "16384 4096 32 20 0 0 40" "16384 4096 32 20 3 0 40" 2>&1 | tee gpurun_out/r4c_ab.txt
