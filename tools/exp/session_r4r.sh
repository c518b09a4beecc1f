#!/bin/bash
# Round 4, last session: every GPU test, smoke, the bench line with its legs, kernel stats of the bench command, the banks.
# (PMC passes: tools/gpu_session.sh — the headline kernel's sources are those of profiles/pmc_traffic.json.)
#   gpurun --timeout 900 -- 'bash tools/exp/session_r4r.sh'
set -u
R=$GRAFT_REPO_ROOT
TAG=r04
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
echo "== pytest -m gpu"; timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (image, all legs)"; timeout 600 python bench.py --steps 50 --warmup 5 2>gpurun_out/bench_image.err | tail -1 | tee gpurun_out/${TAG}_bench_image.json | cut -c1-300
echo "== rocprofv3 kernel stats of the bench command"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --cpu-budget 0 > $R/gpurun_out/prof/stats.log 2>&1 )
python tools/prof_summary.py stats gpurun_out/prof/stats/bench_results.db > gpurun_out/${TAG}_bench_kernel_stats.txt 2>/dev/null; head -8 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-160
echo "== banks"
timeout 300 python tools/bench_firbank.py > gpurun_out/${TAG}_banks.json 2>&1; cut -c1-250 gpurun_out/${TAG}_banks.json
echo "== exact IIR bank: the launches of one call"
( cd /tmp && rm -rf /tmp/iirt && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > gpurun_out/${TAG}_iir_launches.txt 2>&1; tail -4 gpurun_out/${TAG}_iir_launches.txt
rm -rf gpurun_out/prof/stats/*.db
