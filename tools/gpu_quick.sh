#!/bin/bash
# quick perf probe of the N=1024 kernel (production lib, then ablation lib)
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
$B check | tail -1
for run in 8 16 32; do $B bench 1024 512 1 26 0 $run 30 | tail -1; done
$B bench 1024 512 1 26 3 16 30 | tail -1
$B bench 1024 256 1 26 0 16 30 | tail -1
$B bench 16384 8192 64 20 0 0 10 | tail -1
$B bench 4096 1024 16 22 0 0 10 | tail -1
$B bench 256 128 8 24 0 0 10 | tail -1
export LD_LIBRARY_PATH=$R/friture_amd/lib/ablate
for ab in 0 1 2 3 4 12 5 7 15; do echo -n "ablate=$ab: "; FRT_ABLATE=$ab $B bench 1024 512 1 26 0 16 30 | tail -1; done
