#!/bin/bash
# round 5, session n: the headline's runs re-cut into whole rounds of the chip's workgroup slots (variant bx: FRT_STFT_ROUNDS = -1 off, k rounds)
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
S="s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//"
timeout 600 python -m pytest tests/test_stft_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for k in -1 0 2 3 4 5; do
  for cfg in "3 0 40 32 1" "0 0 40 32 1" "3 0 40 32 0"; do
    echo -n "rounds $k: "; FRT_STFT_ROUNDS=$k LD_LIBRARY_PATH=$R/tools/variants/bx:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
  done
done
done
echo "== other shapes (auto vs off): 8 ch x 2^23, hop 256"
for k in -1 0; do
  for cfg in "1024 512 8 23 3 0 40 32 1" "1024 256 1 26 3 0 40 32 1" "1024 512 1 25 3 0 40 64 1"; do
    echo -n "rounds $k: "; FRT_STFT_ROUNDS=$k LD_LIBRARY_PATH=$R/tools/variants/bx:${LD_LIBRARY_PATH:-} timeout 120 $B bench $cfg | tail -1 | sed "s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//"
  done
done
