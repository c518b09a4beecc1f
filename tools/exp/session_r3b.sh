#!/bin/bash
# round 3, session b: wave-uniform arithmetic on the scalar unit, slots as register pairs — parity and A/B against the round-2 build (nomf)
export FRT_BENCH_SETS=4
tools/bin/stft_selftest check | tail -1
for rep in 1 2 3; do
bash tools/exp/ab_variants.sh "base nomf" "1024 512 1 26 3 0 40" "1024 512 1 26 0 0 40"
done
FRT_STFT_RING_IMAGE=1 bash tools/exp/ab_variants.sh "base" "1024 512 1 26 3 0 40"
FRT_STFT_NO_RING=1 bash tools/exp/ab_variants.sh "base" "1024 512 1 26 0 0 40"
bash tools/exp/ab_variants.sh "base nomf" "1024 256 1 26 3 0 40" "1024 256 1 26 0 0 40" "512 256 1 26 3 0 40" "256 128 1 25 0 0 40"
