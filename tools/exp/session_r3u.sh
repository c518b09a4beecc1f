#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_gcc_gpu.py tests/test_install_swap_gpu.py -m gpu -x -q > gpurun_out/r3u_tests.log 2>&1
tail -4 gpurun_out/r3u_tests.log
timeout 600 python tools/stream_latency.py > gpurun_out/r3u_latency.json 2> gpurun_out/r3u_latency.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3u_latency.json"))
for k, v in d.items():
    if "delay" in k: print(f"{k:48s} p50 {v['p50_us']:8.1f}  p99 {v['p99_us']:8.1f}  mean {v['mean_us']:8.1f}")
PY
