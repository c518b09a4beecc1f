#!/bin/bash
# round 3, session e: run lengths chosen for whole rounds of resident workgroups (3 waves/SIMD: 3072 lane groups resident; 4: 4096)
export FRT_BENCH_SETS=4
for run in 43 44 22 23 15 11 16; do
bash tools/exp/ab_variants.sh "base" "1024 512 1 26 3 $run 40" "1024 512 1 26 0 $run 40" | cut -c1-140
done
echo "-- 4 waves/SIMD builds"
for run in 32 33 16 17 11; do
FRT_STFT_RING_IMAGE=1 bash tools/exp/ab_variants.sh "fk3w4" "1024 512 1 26 3 $run 40" | cut -c1-140
bash tools/exp/ab_variants.sh "fk0w4" "1024 512 1 26 0 $run 40" | cut -c1-140
done
