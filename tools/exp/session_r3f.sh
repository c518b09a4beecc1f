#!/bin/bash
# round 3, session f: where the N = 1024 kernel's time goes today (ablation bits: 1 no stores, 2 no loads, 4 no FFT, 8 no unpack shuffles)
export FRT_BENCH_SETS=4
for abl in 0 1 2 3 4 5 6 7 12 15; do
echo -n "ablate=$abl image: "; FRT_ABLATE=$abl bash tools/exp/ab_variants.sh "abl" "1024 512 1 26 3 0 40" | cut -c60-120
echo -n "ablate=$abl psd  : "; FRT_ABLATE=$abl bash tools/exp/ab_variants.sh "abl" "1024 512 1 26 0 0 40" | cut -c60-120
done
echo "no rare path (edge zone 0)"; FRT_EDGE_SCALE=0 bash tools/exp/ab_variants.sh "abl base" "1024 512 1 26 3 0 40" | cut -c1-120
