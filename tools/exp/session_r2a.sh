#!/bin/bash
# round-2 session A: memory-shape variants + NT policy / run-length A/B of the real N=1024 kernel
mkdir -p gpurun_out/r2
tools/bin/membench2 256 4 q > gpurun_out/r2/membench2b.txt 2>&1
grep -E "stft-shape|write|memcpy" gpurun_out/r2/membench2b.txt
for run in 4 8 16 32; do
  bash tools/exp/ab_variants.sh "base nts ntl ntls" "1024 512 1 26 0 $run 40" "1024 512 1 26 3 $run 40"
done 2>&1 | tee gpurun_out/r2/ab_nt.txt
