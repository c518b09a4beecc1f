#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
export FRT_BENCH_SETS=4
for v in base pf2; do
  [ $v = pf2 ] && export LD_LIBRARY_PATH=$R/friture_amd/lib/variants/pf2
  $B check | tail -1
  echo -n "$v psd: "; $B bench 1024 512 1 26 0 0 50 | tail -1
  echo -n "$v img: "; $B bench 1024 512 1 26 3 0 50 | tail -1
  echo -n "$v psd hop256: "; $B bench 1024 256 1 26 0 0 30 | tail -1
  echo -n "$v img N512: "; $B bench 512 256 1 26 3 0 30 | tail -1
done
