#!/bin/bash
for ms in 8 1 2 4; do for wg in 1024 2048 4096; do echo -n "max_slices $ms goal $wg: "; FRT_ZS_MAX_SLICES=$ms FRT_ZS_WAVE_GOAL=$wg python tools/bench_octbank.py --iters 10 | cut -c1-80; done; done
