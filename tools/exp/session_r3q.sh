#!/bin/bash
# round 3, session q: spectrogram object with zero-copy chunk in / pixel block out
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_widgets_gpu.py tests/test_pipeline_gpu.py tests/test_stft_gpu.py -m gpu -x -q > gpurun_out/r3q_tests.log 2>&1
tail -4 gpurun_out/r3q_tests.log
timeout 600 python tools/stream_latency.py > gpurun_out/r3q_latency.json 2> gpurun_out/r3q_latency.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3q_latency.json"))
for k, v in d.items():
    print(f"{k:48s} p50 {v['p50_us']:8.1f}  p99 {v['p99_us']:8.1f}  mean {v['mean_us']:8.1f}")
PY
bash tools/exp/trace_streams.sh specgram 2>&1 | tail -12
