#!/bin/bash
# round 4, session b: where the packed N = 16384 instance spends its frame (s_memtime intervals), two variants, SQ counters
set -u
mkdir -p gpurun_out
for a in "8192 0" "8192 3" "4096 0"; do FRT_LIB_VARIANT=pktime timeout 200 python tools/exp/pk_timing.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4b_timing.txt
echo "== variants"
bash tools/exp/ab_variants.sh "base pktw pkprio base" "16384 8192 32 20 0 0 40" "16384 8192 32 20 3 0 40" "16384 4096 32 20 0 0 40" 2>&1 | tee gpurun_out/r4b_ab.txt
echo "== counters"
bash tools/gpu_pmc.sh r4b_pk 0 0 16384 8192 32 20 sq > /dev/null 2>&1; python tools/prof_summary.py pmc gpurun_out/pmc_r4b_pk stft_pk | tee gpurun_out/r4b_pk_pmc.txt
