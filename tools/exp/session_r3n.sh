#!/bin/bash
# round 3, session n: the delay estimator's C object (tests + chunk latency + device time per chunk)
mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_gcc_gpu.py -m gpu -x -q > gpurun_out/r3n_tests.log 2>&1
tail -3 gpurun_out/r3n_tests.log
timeout 600 python tools/stream_latency.py > gpurun_out/r3n_latency.json 2> gpurun_out/r3n_latency.err
grep -A5 '"delay_' gpurun_out/r3n_latency.json | head -60
cat > /tmp/dp.py <<'PY'
import numpy as np, sys
sys.path.insert(0, "/root/repo")
from friture_amd.delay_estimator import DelayEstimatorStream
b = DelayEstimatorStream(1.0)
x = np.random.default_rng(0).standard_normal((2, 512))
for i in range(300):
    b.handle_new_data(x)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/prof_delay -o d -- python /tmp/dp.py > /dev/null 2>&1
cd - > /dev/null
f=$(find /tmp/prof_delay -name '*kernel_stats.csv' | head -1); head -8 "$f"
f=$(find /tmp/prof_delay -name '*memory_copy_stats.csv' | head -1); head -5 "$f"
