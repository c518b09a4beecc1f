#!/bin/bash
# round 5, session m: non-temporal row stores in the large-frame family, by hop: none / hop N/4 only (shipped) / both hops
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "large_frame or all_sizes" 2>&1 | tail -2
for cfg in "8192 4096 32 21" "8192 2048 32 21" "4096 2048 16 22" "4096 1024 16 22" "2048 1024 8 24" "2048 512 8 24" "16384 8192 32 20" "16384 4096 32 20"; do
 for kind in 0 3; do
  for v in pknone base pkboth; do
    if [ $v = base ]; then LP=""; else LP=$R/tools/variants/$v; fi
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench $cfg $kind 0 40 | tail -1 | sed "s/bench p32 //; s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//; s/ run=0 sets=4 packed//"
  done
 done
done
