#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
$B check | tail -1
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for v in r1head base; do
  if [ $v = base ]; then LP=""; else LP=$PWD/tools/variants/$v; fi
  for cfg in "1024 512 1 26 3" "2048 1024 8 24 3" "4096 1024 16 22 3" "8192 4096 32 21 3" "16384 8192 32 20 3"; do echo -n "$v $cfg: "; LD_LIBRARY_PATH=$LP $B bench $cfg 0 40 | tail -1 | cut -c65-90; done
done
done
