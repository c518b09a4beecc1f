#!/bin/bash
# round 4, session j: four consecutive bins per thread, 16-byte row stores
set -u
mkdir -p gpurun_out
echo "== parity"; timeout 300 python tools/exp/pk_debug.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q -m gpu -k "large_frame or lds_staged or image_epilogue or db_norm or all_sizes or randomised" -p no:cacheprovider 2>&1 | tail -3
for a in "8192 0" "8192 3"; do FRT_LIB_VARIANT=pktime timeout 200 python tools/exp/pk_timing.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4j_timing.txt
echo "== bench"
export FRT_BENCH_SETS=4
for rep in 1 2; do for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3"; do tools/bin/stft_selftest bench $cfg 0 40 | tail -1 | cut -c1-175; done; done | tee gpurun_out/r4j_ab.txt
