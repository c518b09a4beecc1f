#!/bin/bash
# Quick look on a GPU box (about a minute): self-test, cold-input rates of the main STFT sizes, octave-bank leg.
#   gpurun --timeout 600 -- 'bash tools/gpu_quick.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=$R/tools/bin/stft_selftest
export FRT_BENCH_SETS=4                       # rotate over four buffer sets: no launch finds its input in the Infinity Cache
$B check | tail -1
for cfg in "1024 512 1 26" "1024 256 1 26" "16384 8192 32 20" "4096 1024 16 22"; do
  set -- $cfg
  echo -n "psd: "; $B bench $1 $2 $3 $4 0 0 30 | tail -1
  echo -n "img: "; $B bench $1 $2 $3 $4 3 0 30 | tail -1
done
cd $R && python tools/bench_octbank.py 2>/dev/null | tail -1
