#!/bin/bash
mkdir -p gpurun_out/r2
tools/bin/stft_selftest check | tail -2
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q 2>&1 | tail -15
bash tools/exp/ab_variants.sh "r1head base" "1024 512 1 26 0 0 40" "1024 512 1 26 3 0 40" "1024 256 1 26 3 0 40" "2048 1024 8 24 3 0 30" "4096 1024 16 22 3 0 30" "16384 8192 32 20 3 0 30" 2>&1 | tee gpurun_out/r2/ab_exact.txt
