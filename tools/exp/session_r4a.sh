#!/bin/bash
# round 4, session a: the packed N = 16384 instance — parity, then A/B against round 3's instance
set -u
mkdir -p gpurun_out
echo "== pk_debug"; timeout 300 python tools/exp/pk_debug.py 2>&1 | tail -40
echo "== large-frame tests"; timeout 900 python -m pytest tests/test_stft_gpu.py -x -q -m gpu -k "large_frame or all_sizes or lds_staged or randomised" -p no:cacheprovider 2>&1 | tail -5
echo "== A/B (stft_selftest bench N hop C log2T kind)"
export FRT_BENCH_SETS=4
for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3"; do
  for v in pk old; do
    if [ $v = old ]; then export FRT_STFT_NO_PK=1; else unset FRT_STFT_NO_PK; fi
    echo "$v: $(tools/bin/stft_selftest bench $cfg 0 40 | tail -1)"
  done
done 2>&1 | tee gpurun_out/r4a_ab.txt
unset FRT_STFT_NO_PK
