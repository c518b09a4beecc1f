#!/bin/bash
for rep in 1 2 3; do for v in "" vout; do echo -n "${v:-base}: "; FRT_LIB_VARIANT=$v python tools/bench_octbank.py --iters 10 | cut -c1-80; done; done
