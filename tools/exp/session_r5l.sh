#!/bin/bash
# round 5, session l: non-temporal row stores in stft_pk16_kernel (N = 16384; rows of 8193 values start on 4-byte boundaries)
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
for rep in 1 2; do
for v in base pknt; do
  if [ $v = base ]; then LP=""; else LP=$R/tools/variants/$v; fi
  for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3"; do
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench $cfg 0 40 | tail -1 | sed "s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//"
  done
done
done
