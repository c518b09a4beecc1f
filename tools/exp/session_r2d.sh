#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
for sc in 1 0 0.3 3; do
  echo -n "edge_scale=$sc: "; FRT_EDGE_SCALE=$sc $B bench 1024 512 1 26 3 0 40 | tail -1 | sed 's/bench N=//; s/T=2^[0-9]* //; s/sets=4: //; s/spectra.s.*GB.s//'
done
echo -n "psd: "; $B bench 1024 512 1 26 0 0 40 | tail -1
echo -n "r1head img: "; LD_LIBRARY_PATH=$PWD/tools/variants/r1head $B bench 1024 512 1 26 3 0 40 | tail -1
echo -n "r1head psd: "; LD_LIBRARY_PATH=$PWD/tools/variants/r1head $B bench 1024 512 1 26 0 0 40 | tail -1
