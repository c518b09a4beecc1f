#!/bin/bash
# round 4, session n: zero-state kernel knobs (waves per launch, K-slices) on the configs[2] batch
for env in "" "FRT_ZS_WAVE_GOAL=4096" "FRT_ZS_WAVE_GOAL=8192" "FRT_ZS_WAVE_GOAL=1024" "FRT_ZS_MAX_SLICES=1" ""; do
  echo "${env:-default}: $(env $env python tools/bench_octbank.py --bpo 3 --log2-samples 22 --channels 8 --chunk 1024 --iters 10 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.4f ms"%r["ms"])')  bpo24: $(env $env python tools/bench_octbank.py --bpo 24 --log2-samples 20 --channels 8 --chunk 1024 --iters 10 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.4f ms"%r["ms"])')"
done
for chunk in 2048 4096; do echo "chunk $chunk: $(python tools/bench_octbank.py --bpo 3 --log2-samples 22 --channels 8 --chunk $chunk --iters 10 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.4f ms"%r["ms"])')"; done
