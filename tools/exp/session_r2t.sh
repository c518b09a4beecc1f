#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
LD_LIBRARY_PATH=$PWD/tools/variants/pinp:$LD_LIBRARY_PATH timeout 300 $B check 2>&1 | tail -1
for rep in 1 2 3 4; do
  echo -n "base ring psd: "; $B bench 1024 512 1 26 0 0 100 | tail -1 | cut -c60-130
  echo -n "pinp ring psd: "; LD_LIBRARY_PATH=$PWD/tools/variants/pinp:$LD_LIBRARY_PATH $B bench 1024 512 1 26 0 0 100 | tail -1 | cut -c60-130
  echo -n "base win  img: "; $B bench 1024 512 1 26 3 0 100 | tail -1 | cut -c60-130
  echo -n "pinp win  img: "; LD_LIBRARY_PATH=$PWD/tools/variants/pinp:$LD_LIBRARY_PATH $B bench 1024 512 1 26 3 0 100 | tail -1 | cut -c60-130
done
