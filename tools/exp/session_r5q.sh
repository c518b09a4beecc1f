#!/bin/bash
# round 5, session q: where stft_pk16r_kernel's time goes — ablation builds (FRT_PKR_ABLATE: 1 no window loads, 2 no unpack-factor loads,
# 4 no second twiddle products, 8 no row stores, 16 no sample loads after a run's first frame; wrong output by design), colour and PSD
# kinds at hop N/2; and the full-size shard test's message
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
S="s/algorithmic.*of 8 TB.s)//; s/bench p32 N=16384 //"
echo "== full-size shard test"
timeout 300 python -m pytest tests/test_stft_gpu.py -x -q -k "large_frame_shard_full_size" 2>&1 | grep -v "^$" | tail -30
echo "== ablations, two rounds"
for rep in 1 2; do
  for v in px a1 a2 a4 a8 a16 a17 a27; do
    for cfg in "16384 8192 32 20 3" "16384 8192 32 20 0"; do
      echo -n "$v: "; LD_LIBRARY_PATH=$R/tools/variants/$v:${LD_LIBRARY_PATH:-} timeout 120 $B bench $cfg 0 40 | tail -1 | sed "$S"
    done
  done
  echo -n "pk16: "; FRT_STFT_NO_PK16R=1 LD_LIBRARY_PATH=$R/tools/variants/px timeout 120 $B bench 16384 8192 32 20 3 0 40 | tail -1 | sed "$S"
  echo -n "pk16: "; FRT_STFT_NO_PK16R=1 LD_LIBRARY_PATH=$R/tools/variants/px timeout 120 $B bench 16384 8192 32 20 0 0 40 | tail -1 | sed "$S"
done
