#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
for st in 0 1 2 4 8; do
  for cfg in "16384 8192 32 20 0" "8192 4096 32 21 0" "4096 1024 16 22 0" "2048 1024 8 24 0"; do echo -n "stagger=$st $cfg: "; FRT_BIG_STAGGER=$st $B bench $cfg 0 40 | tail -1 | cut -c60-95; done
done
