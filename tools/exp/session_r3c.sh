#!/bin/bash
# round 3, session c: output kind as a compile-time constant, 3 vs 4 waves per SIMD, window vs ring instance
export FRT_BENCH_SETS=4
for rep in 1 2; do
echo "-- image, window instance"; bash tools/exp/ab_variants.sh "base fk3w3 fk3w4" "1024 512 1 26 3 0 40"
echo "-- image, ring instance"; FRT_STFT_RING_IMAGE=1 bash tools/exp/ab_variants.sh "base fk3w3 fk3w4" "1024 512 1 26 3 0 40"
echo "-- psd, ring instance"; bash tools/exp/ab_variants.sh "base fk0w3 fk0w4" "1024 512 1 26 0 0 40"
echo "-- psd, window instance"; FRT_STFT_NO_RING=1 bash tools/exp/ab_variants.sh "base fk0w3 fk0w4" "1024 512 1 26 0 0 40"
done
echo "-- hop 256"; bash tools/exp/ab_variants.sh "base fk3w4 fk0w4" "1024 256 1 26 3 0 40" "1024 256 1 26 0 0 40"
