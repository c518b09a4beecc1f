#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
$B check | grep -E "FAIL|SELFTEST"
echo "== generic"; $B bench 1024 512 1 26 0 -11 30 | tail -1; $B bench 1024 512 1 26 3 -16 30 | tail -1
echo "== wave (4 waves/SIMD bound)"; for kind in 0 3; do for run in 8 11 16 21 32; do $B bench 1024 512 1 26 $kind $run 30 | tail -1; done; done
$B bench 1024 256 1 26 0 16 30 | tail -1
$B bench 256 128 8 24 0 0 10 | tail -1
export LD_LIBRARY_PATH=$R/friture_amd/lib/variants/nobound; echo "== wave, no bound"; for kind in 0 3; do for run in 11 16; do $B bench 1024 512 1 26 $kind $run 30 | tail -1; done; done
