#!/bin/bash
# A/B of library variants built by build_variant.sh (LD_LIBRARY_PATH beats the driver's RUNPATH).
#   bash tools/exp/ab_variants.sh "<variants>" "<bench args>" ["<bench args>" ...]
B=tools/bin/stft_selftest; export FRT_BENCH_SETS=4
VARS=$1; shift
for v in $VARS; do
  if [ $v = base ]; then LP=""; else LP=$PWD/tools/variants/$v; fi
  for cfg in "$@"; do
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH $B bench $cfg | tail -1 | sed "s/algorithmic.*of 8 TB.s)//"
  done
done
