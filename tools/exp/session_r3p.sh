#!/bin/bash
# round 3, session p: the octave bank's chunk path (two launches)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ola_gpu.py tests/test_widgets_gpu.py tests/test_iir_gpu.py -m gpu -x -q > gpurun_out/r3p_tests.log 2>&1
tail -15 gpurun_out/r3p_tests.log
timeout 600 python tools/stream_latency.py > gpurun_out/r3p_latency.json 2> gpurun_out/r3p_latency.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3p_latency.json"))
for k, v in d.items():
    if "octave" in k:
        print(f"{k:48s} p50 {v['p50_us']:8.1f}  p99 {v['p99_us']:8.1f}  mean {v['mean_us']:8.1f}")
PY
bash tools/exp/trace_streams.sh octave 2>&1 | tail -24
