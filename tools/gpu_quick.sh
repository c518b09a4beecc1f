#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
for cfg in "16384 8192 32 20" "16384 8192 2 24" "8192 4096 32 21" "4096 2048 16 22" "2048 1024 16 22" "2048 1024 1 26"; do
  set -- $cfg
  for r in 1 2 3 4 8 16; do
  echo -n "big r$r:  "; $B bench $1 $2 $3 $4 0 $r 10 | tail -1
  done
done
