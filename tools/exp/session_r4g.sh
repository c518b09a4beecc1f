#!/bin/bash
set -u
mkdir -p gpurun_out
bash tools/exp/ab_variants.sh "base pktw base pktw" "16384 8192 32 20 0 0 40" "16384 8192 32 20 3 0 40" "16384 4096 32 20 0 0 40" "16384 4096 32 20 3 0 40" 2>&1 | tee gpurun_out/r4g_ab.txt
echo "== run lengths, hop 4096"
export FRT_BENCH_SETS=4
for run in 16 22 32 43 64; do tools/bin/stft_selftest bench 16384 4096 32 20 0 $run 40 | tail -1 | cut -c1-160; done
for run in 8 16 22 32; do tools/bin/stft_selftest bench 16384 8192 32 20 0 $run 40 | tail -1 | cut -c1-160; done
