#!/bin/bash
# round 3, session h: cache-policy bits on the row stores of the N = 1024 kernel
export FRT_BENCH_SETS=4
LD_LIBRARY_PATH=$PWD/tools/variants/st_sc1 tools/bin/stft_selftest check | tail -1
for rep in 1 2; do
bash tools/exp/ab_variants.sh "base st_sc1 st_sc0 st_sc0sc1 st_sc1nt st_sc0sc1nt" "1024 512 1 26 3 0 40" "1024 512 1 26 0 0 40" | cut -c1-130
done
