#!/bin/bash
export FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
for rep in 1 2 3 4 5 6; do
  echo -n "window img: "; FRT_STFT_NO_RING=1 $B bench 1024 512 1 26 3 0 100 | tail -1 | cut -c60-100
  echo -n "ring   img: "; FRT_STFT_RING_IMAGE=1 $B bench 1024 512 1 26 3 0 100 | tail -1 | cut -c60-100
done
for rep in 1 2 3; do
FRT_STFT_NO_RING=1 python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench window', r['roofline']['kernel_ms_repeats'])"
FRT_STFT_RING_IMAGE=1 python bench.py --steps 50 --warmup 5 --cpu-budget 0 --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('bench ring  ', r['roofline']['kernel_ms_repeats'])"
done
