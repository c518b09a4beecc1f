#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
export FRT_BENCH_SETS=4
$B check | tail -1
for cfg in "16384 8192 32 20" "8192 4096 32 21" "4096 2048 16 22" "2048 1024 16 22"; do
  set -- $cfg
  echo -n "psd: "; $B bench $1 $2 $3 $4 0 0 10 | tail -1
  echo -n "img: "; $B bench $1 $2 $3 $4 3 0 10 | tail -1
done
cd $R; python -m pytest tests/test_stft_gpu.py -x -q 2>&1 | tail -2
