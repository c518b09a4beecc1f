#!/bin/bash
# round 4, session k: band filters per wavefront of the IIR lane kernel (occupancy: 1024 waves on 1024 SIMDs at 3 per wave)
for v in "" lane2 lane1 "" lane1; do
  for cfg in "--bpo 3 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 24 --log2-samples 20 --channels 8 --chunk 1024"; do
    echo "${v:-base} $cfg: $(FRT_LIB_VARIANT=$v python tools/bench_octbank.py $cfg --iters 10 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.4f ms  %.3e octave-bands/s"%(r["ms"],r["octave_bands_per_s"]))')"
  done
done
