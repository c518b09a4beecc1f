#!/bin/bash
# round 3, session m: is the store-less build of the N = 1024 kernel latency bound?  3 vs 4 waves per SIMD (colour kind as a compile-time
# constant, ring instance: 122 VGPRs), with and without the row stores
export FRT_BENCH_SETS=4 FRT_STFT_RING_IMAGE=1
for abl in 0 1; do
for v in ab3 ab4; do echo -n "ablate=$abl "; FRT_ABLATE=$abl bash tools/exp/ab_variants.sh "$v" "1024 512 1 26 3 0 40" | cut -c1-120; done
done
