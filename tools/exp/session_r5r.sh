#!/bin/bash
# round 5, session r: stft_pk16r_kernel — (1) where its output differs from stft_pk16_kernel's on the full-size shard (tools/exp/pkr_debug.py),
# (2) the second half of the launch's workgroups started late (FRT_PKR_STAGGER = 28 / 55 / 83 sleeps of 64 cycles): do two workgroups of a
# CU that stall in lockstep hide each other's memory phases when they run half a frame apart?
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
S="s/algorithmic.*of 8 TB.s)//; s/bench p32 N=16384 //"
echo "== (1) pk16r against pk16, full-size shard"
FRT_LIB_VARIANT=px timeout 300 python tools/exp/pkr_debug.py 8192 2>&1 | grep -v amdgpu.ids | tail -30
echo "== (2) stagger, two rounds"
for rep in 1 2; do
  for v in px s28 s55 s83; do
    for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3"; do
      echo -n "$v: "; LD_LIBRARY_PATH=$R/tools/variants/$v:${LD_LIBRARY_PATH:-} timeout 120 $B bench $cfg 0 40 | tail -1 | sed "$S"
    done
  done
done
