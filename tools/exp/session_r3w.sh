#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ola_gpu.py tests/test_widgets_gpu.py tests/test_install_swap_gpu.py tests/test_upstream_properties.py -m gpu -x -q > gpurun_out/r3w_tests.log 2>&1
tail -6 gpurun_out/r3w_tests.log
timeout 600 python tools/stream_latency.py > gpurun_out/r3w_latency.json 2> gpurun_out/r3w_latency.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3w_latency.json"))
for k, v in d.items():
    if "octave" in k: print(f"{k:48s} p50 {v['p50_us']:8.1f}  p99 {v['p99_us']:8.1f}  mean {v['mean_us']:8.1f}")
PY
