#!/bin/bash
for env in "FRT_ZS_MAX_SLICES=1" "FRT_ZS_WAVE_GOAL=256" "FRT_ZS_WAVE_GOAL=512" "FRT_ZS_MAX_SLICES=2" "FRT_ZS_MAX_SLICES=1"; do
  for cfg in "--bpo 3 --log2-samples 22 --channels 8" "--bpo 24 --log2-samples 20 --channels 8" "--bpo 3 --log2-samples 18 --channels 1" "--bpo 24 --log2-samples 20 --channels 64"; do
    echo "$env $cfg: $(env $env python tools/bench_octbank.py $cfg --chunk 1024 --iters 10 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.4f ms"%r["ms"])')"
  done
done
