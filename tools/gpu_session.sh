#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats + PMC traffic.  Outputs -> gpurun_out/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== selftest"; timeout 300 tools/bin/stft_selftest check | tail -2
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (image)"; timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_image.json
echo "== bench (psd)"; timeout 600 python bench.py --steps 50 --warmup 5 --kind psd --cpu-budget 0 2>&1 | tail -1 | tee gpurun_out/bench_psd.json
echo "== rocprofv3 kernel stats of the bench command"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --cpu-budget 0 > $R/gpurun_out/prof/stats.log 2>&1 )
echo "== rocprofv3 PMC passes of the bench command (separate passes, kernel-trace only)"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/prof/pmc_bench/$n -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --prewarm-ms 0 > $R/gpurun_out/prof/pmc_$n.log 2>&1 ); echo "pmc $n rc=$?"
done
ls gpurun_out/prof/stats | tail -2
