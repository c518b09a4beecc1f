#!/bin/bash
timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_widgets_gpu.py tests/test_sharding_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do python tools/bench_octbank.py --iters 10 | cut -c1-80; FRT_IIR_EXACT_OPS=1 python tools/bench_octbank.py --iters 10 | cut -c1-80; done
python tools/bench_octbank.py --iters 5 --chunk 4096 --bpo 24 --log2-samples 20 | cut -c1-80
