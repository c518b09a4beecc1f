#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -4
for ch in 4096 2048 -2048; do python tools/bench_octbank.py --chunk $ch 2>/dev/null | tail -1 | cut -c1-200; done
python tools/bench_octbank.py --chunk 4096 --channels 64 --bpo 24 --log2-samples 20 2>/dev/null | tail -1 | cut -c1-200
