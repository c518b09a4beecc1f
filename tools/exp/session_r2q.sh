#!/bin/bash
for ch in 1024 2048 4096; do echo -n "bpo3 chunk $ch: "; python tools/bench_octbank.py --iters 10 --chunk $ch | cut -c1-70; done
for ch in 2048 4096 8192; do echo -n "bpo24 chunk $ch: "; python tools/bench_octbank.py --iters 5 --chunk $ch --bpo 24 --log2-samples 20 | cut -c1-70; done
for ch in 2048 4096 8192; do echo -n "bpo24 64ch chunk $ch: "; python tools/bench_octbank.py --iters 3 --chunk $ch --bpo 24 --log2-samples 20 --channels 64 | cut -c1-70; done
