#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
export FRT_BENCH_SETS=3
for v in base nolut nolog; do
  [ $v != base ] && export LD_LIBRARY_PATH=$R/friture_amd/lib/variants/$v
  for rep in 1 2; do
  echo -n "$v psd: "; $B bench 1024 512 1 26 0 0 50 | tail -1
  echo -n "$v img: "; $B bench 1024 512 1 26 3 0 50 | tail -1
  done
done
