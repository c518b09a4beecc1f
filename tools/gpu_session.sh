#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats.  Outputs -> gpurun_out/.
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== selftest"; timeout 300 tools/bin/stft_selftest check | tail -3
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench (image)"; timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -2 | tee gpurun_out/bench_image.json
echo "== bench (psd)"; timeout 600 python bench.py --steps 50 --warmup 5 --kind psd --cpu-budget 0 2>&1 | tail -1 | tee gpurun_out/bench_psd.json
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-budget 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/stats.log 2>&1 )
find gpurun_out/prof/stats -name "*kernel_stats*" | head; f=$(find gpurun_out/prof/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
