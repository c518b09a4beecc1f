#!/bin/bash
# soak: the kernels whose waits are hand-counted (LDS-DMA instances), repeated, under concurrent load from a second process
( for i in 1 2 3 4 5 6; do python tools/bench_firbank.py > /dev/null 2>&1; done ) &
BG=$!
fail=0
for i in $(seq 1 12); do
  timeout 300 python -m pytest tests/test_stft_gpu.py tests/test_capi_selftest_gpu.py -x -q -m gpu -k "ring or large_frame or image_epilogue or selftest or randomised" -p no:cacheprovider 2>&1 | tail -1 | grep -q "passed" || { fail=$((fail+1)); echo "iteration $i FAILED"; }
done
wait $BG
echo "soak done, failures: $fail"
