#!/bin/bash
# round 5, session k: the ring instance for the colour kind on split rows with non-temporal stores (variant bx, FRT_STFT_RING_IMAGE=1)
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
S="s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//"
for rep in 1 2 3; do
  for mode in window ring; do
    if [ $mode = ring ]; then export FRT_STFT_RING_IMAGE=1; else unset FRT_STFT_RING_IMAGE; fi
    for cfg in "3 0 40 32 1" "3 12 40 32 1" "3 20 40 32 1"; do
      echo -n "$mode: "; LD_LIBRARY_PATH=$R/tools/variants/bx:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
    done
  done
done
unset FRT_STFT_RING_IMAGE
cp friture_amd/lib/libfriture_hip.so /tmp/base.so; cp tools/variants/bx/libfriture_hip.so friture_amd/lib/libfriture_hip.so
FRT_STFT_RING_IMAGE=1 timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "image or split or ring" 2>&1 | tail -2
cp /tmp/base.so friture_amd/lib/libfriture_hip.so
