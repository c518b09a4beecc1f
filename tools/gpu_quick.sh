#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
$B check | grep -E "FAIL|SELFTEST"
for kind in 0 3; do for run in 8 11 16 21 32; do $B bench 1024 512 1 26 $kind $run 30 | tail -1; done; done
$B bench 1024 256 1 26 0 16 30 | tail -1
$B bench 256 128 8 24 0 0 10 | tail -1
$B bench 512 256 4 25 3 0 10 | tail -1
$B bench 16384 8192 64 20 0 0 10 | tail -1
$B bench 4096 1024 16 22 0 0 10 | tail -1
