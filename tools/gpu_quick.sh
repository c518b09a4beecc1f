#!/bin/bash
R=$GRAFT_REPO_ROOT
B=$R/tools/bin/stft_selftest
export FRT_BENCH_SETS=4
$B check | grep -E "pix_bad=[1-9]|FAIL|SELFTEST" | head -5
echo -n "fast img: "; $B bench 1024 512 1 26 3 0 50 | tail -1
echo -n "fast img: "; $B bench 1024 512 1 26 3 0 50 | tail -1
echo -n "     psd: "; $B bench 1024 512 1 26 0 0 50 | tail -1
export FRT_IMAGE_EXACT_EPS=1
echo -n "eps  img: "; $B bench 1024 512 1 26 3 0 50 | tail -1
echo -n "eps  img: "; $B bench 1024 512 1 26 3 0 50 | tail -1
unset FRT_IMAGE_EXACT_EPS
cd $R && timeout 600 python -m pytest tests/test_stft_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -3
