#!/bin/bash
timeout 600 python -m pytest tests/test_ola_gpu.py tests/test_widgets_gpu.py -x -q -m gpu 2>&1 | tail -3
for cfg in "8 3 22 10" "8 24 20 5"; do python tools/exp/fir_only.py $cfg; done
