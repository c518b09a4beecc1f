#!/bin/bash
for rep in 1 2; do
bash tools/exp/ab_variants.sh "base unr" "16384 8192 32 20 0 0 40" "16384 8192 32 20 3 0 40" "8192 4096 32 21 0 0 40" "4096 2048 16 22 0 0 40" "4096 1024 16 22 3 0 40"
done
