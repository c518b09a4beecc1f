#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
LD_LIBRARY_PATH=$PWD/tools/variants/ringh3:$LD_LIBRARY_PATH timeout 300 $B check 2>&1 | grep -v "^ok" | tail -3
for rep in 1 2 3; do
  echo -n "ring2 psd: "; $B bench 1024 512 1 26 0 0 50 | tail -1 | cut -c60-130
  echo -n "ring2 img: "; $B bench 1024 512 1 26 3 0 50 | tail -1 | cut -c60-130
  echo -n "ring3h psd: "; LD_LIBRARY_PATH=$PWD/tools/variants/ringh3:$LD_LIBRARY_PATH $B bench 1024 512 1 26 0 0 50 | tail -1 | cut -c60-130
  echo -n "ring3h img: "; LD_LIBRARY_PATH=$PWD/tools/variants/ringh3:$LD_LIBRARY_PATH $B bench 1024 512 1 26 3 0 50 | tail -1 | cut -c60-130
  echo -n "window psd: "; FRT_STFT_NO_RING=1 $B bench 1024 512 1 26 0 0 50 | tail -1 | cut -c60-130
  echo -n "window img: "; FRT_STFT_NO_RING=1 $B bench 1024 512 1 26 3 0 50 | tail -1 | cut -c60-130
done
