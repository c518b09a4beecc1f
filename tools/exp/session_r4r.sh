#!/bin/bash
# refresh of the large-frame evidence after stft_pk16q_kernel became the N = 4096 instance
R=$GRAFT_REPO_ROOT; TAG=r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1
( export FRT_BENCH_SETS=4; for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "8192 4096 32 21 0" "8192 4096 32 21 3" "4096 2048 16 22 0" "4096 1024 16 22 3" "2048 1024 8 24 0" "2048 512 8 24 3"; do tools/bin/stft_selftest bench $cfg 0 40 | tail -1; done ) > gpurun_out/${TAG}_stft_big_bench.txt 2>&1; cat gpurun_out/${TAG}_stft_big_bench.txt | cut -c1-150
bash tools/gpu_pmc.sh ${TAG}_n4096 0 3 4096 1024 16 22 > /dev/null 2>&1; python tools/prof_summary.py pmc gpurun_out/pmc_${TAG}_n4096 stft_pk16q > gpurun_out/${TAG}_stft4096_pmc.txt; head -12 gpurun_out/${TAG}_stft4096_pmc.txt | cut -c1-110
