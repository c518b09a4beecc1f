#!/bin/bash
for sets in 1 2 3 4 6; do FRT_BENCH_SETS=$sets tools/bin/stft_selftest bench 1024 512 1 26 0 0 50 | tail -1 | cut -c55-160; done
for nb in 1 2 3 4 6; do python bench.py --steps 50 --warmup 5 --kind psd --cpu-budget 0 --no-legs --batches $nb 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('psd batches', $nb, r['ms_per_step'], r['roofline']['kernel_ms_repeats'], r['roofline']['frac'])"; done
