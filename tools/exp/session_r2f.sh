#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
$B check | tail -1
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for k in 0 3; do
  echo -n "r1head kind=$k: "; LD_LIBRARY_PATH=$PWD/tools/variants/r1head $B bench 1024 512 1 26 $k 0 60 | tail -1 | cut -c65-90
  echo -n "stream kind=$k: "; $B bench 1024 512 1 26 $k 0 60 | tail -1 | cut -c65-90
  echo -n "nostream kind=$k: "; FRT_STFT_NO_STREAM=1 $B bench 1024 512 1 26 $k 0 60 | tail -1 | cut -c65-90
done
done
for run in 8 11 16 22 32 43; do echo -n "stream img run=$run: "; $B bench 1024 512 1 26 3 $run 60 | tail -1 | cut -c65-90; done
echo -n "stream hop256 img: "; $B bench 1024 256 1 26 3 0 40 | tail -1 | cut -c65-90
echo -n "nostream hop256 img: "; FRT_STFT_NO_STREAM=1 $B bench 1024 256 1 26 3 0 40 | tail -1 | cut -c65-90
