#!/bin/bash
mkdir -p gpurun_out/r2
for run in 43 44 22 21 15 11 16; do
  bash tools/exp/ab_variants.sh "base nts" "1024 512 1 26 0 $run 40" "1024 512 1 26 3 $run 40"
done 2>&1 | tee gpurun_out/r2/ab_run.txt
