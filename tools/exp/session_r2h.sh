#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
$B check | tail -1
timeout 900 python -m pytest tests/test_stft_gpu.py tests/test_sharding_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
for v in r1head base; do
  if [ $v = base ]; then LP=""; else LP=$PWD/tools/variants/$v; fi
  for cfg in "1024 512 1 26 3" "1024 512 1 26 0"; do echo -n "$v $cfg: "; LD_LIBRARY_PATH=$LP $B bench $cfg 0 60 | tail -1 | cut -c65-90; done
done
done
for v in r1head base; do
  if [ $v = base ]; then LP=""; else LP=$PWD/tools/variants/$v; fi
  for cfg in "1024 256 1 26 3" "1024 256 1 26 0" "512 256 2 25 3" "256 128 4 24 0" "1024 384 1 26 0"; do echo -n "$v $cfg: "; LD_LIBRARY_PATH=$LP $B bench $cfg 0 40 | tail -1 | cut -c65-90; done
done
