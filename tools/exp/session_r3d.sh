#!/bin/bash
# round 3, session d: run-length sweep of the N = 1024 kernel (memory-side locality)
export FRT_BENCH_SETS=4
for run in 2 4 6 8 12 16 32; do
bash tools/exp/ab_variants.sh "base" "1024 512 1 26 3 $run 40" "1024 512 1 26 0 $run 40" | cut -c1-140
done
echo "-- 4 waves/SIMD builds"
for run in 2 4 8; do
FRT_STFT_RING_IMAGE=1 bash tools/exp/ab_variants.sh "fk3w4" "1024 512 1 26 3 $run 40" | cut -c1-140
bash tools/exp/ab_variants.sh "fk0w4" "1024 512 1 26 0 $run 40" | cut -c1-140
done
