#!/bin/bash
# round 3, session i: where the N = 16384 kernel's time goes today (FRT_BIG_ABLATE bits: 1 no row stores, 2 no sample loads after the first frame, 4 no sub-transforms)
export FRT_BENCH_SETS=4
for kind in 0 3; do
bash tools/exp/ab_variants.sh "base babl1 babl2 babl3 babl4 babl7" "16384 8192 32 20 $kind 0 40" | cut -c1-150
done
bash tools/exp/ab_variants.sh "base babl1 babl2 babl3 babl4 babl7" "16384 4096 32 20 0 0 40" "4096 1024 16 22 0 0 40" | cut -c1-150
