#!/bin/bash
# round 3, session j: lane-per-chunk output pass of the exact IIR bank (energy-only, time-parallel) vs the slot kernel
timeout 600 python -m pytest tests/test_iir_gpu.py -x -q -m gpu 2>&1 | tail -3
for chunk in 2048 1024 512 256; do
echo -n "lane  bpo 3 chunk $chunk: "; python tools/bench_octbank.py --chunk $chunk 2>/dev/null | tail -1 | cut -c1-120
done
echo -n "slot  bpo 3 chunk 2048: "; FRT_IIR_NO_LANE_KERNEL=1 python tools/bench_octbank.py --chunk 2048 2>/dev/null | tail -1 | cut -c1-120
for chunk in 4096 2048 1024 512; do
echo -n "lane  bpo 24 chunk $chunk: "; python tools/bench_octbank.py --bpo 24 --log2-samples 20 --chunk $chunk 2>/dev/null | tail -1 | cut -c1-120
done
echo -n "slot  bpo 24 chunk 4096: "; FRT_IIR_NO_LANE_KERNEL=1 python tools/bench_octbank.py --bpo 24 --log2-samples 20 --chunk 4096 2>/dev/null | tail -1 | cut -c1-120
echo -n "lane  bpo 24 64ch chunk 2048: "; python tools/bench_octbank.py --bpo 24 --channels 64 --log2-samples 20 --chunk 2048 2>/dev/null | tail -1 | cut -c1-120
echo -n "slot  bpo 24 64ch chunk 4096: "; FRT_IIR_NO_LANE_KERNEL=1 python tools/bench_octbank.py --bpo 24 --channels 64 --log2-samples 20 --chunk 4096 2>/dev/null | tail -1 | cut -c1-120
