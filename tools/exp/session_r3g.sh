#!/bin/bash
# round 3, session g: per-wavefront cycle probes of the N = 1024 kernel
export FRT_BENCH_SETS=4
for abl in 0 1 6; do
for kind in 3 0; do
echo "== ablate=$abl kind=$kind window instance"; FRT_STFT_NO_RING=1 FRT_ABLATE=$abl LD_LIBRARY_PATH=$PWD/tools/variants/probe tools/bin/stft_selftest bench 1024 512 1 26 $kind 0 40 2>&1 | tail -3 | cut -c1-230
done
done
echo "== ring instance, psd"; LD_LIBRARY_PATH=$PWD/tools/variants/probe tools/bin/stft_selftest bench 1024 512 1 26 0 0 40 2>&1 | tail -2 | cut -c1-230
echo "== ring instance, image"; FRT_STFT_RING_IMAGE=1 LD_LIBRARY_PATH=$PWD/tools/variants/probe tools/bin/stft_selftest bench 1024 512 1 26 3 0 40 2>&1 | tail -2 | cut -c1-230
