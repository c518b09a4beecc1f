#!/bin/bash
# round 6, session p: after the quad-row scan — whole GPU suite, traffic of the two exact-bank legs, launch lists, bench
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_leg_traffic.sh r6p_traffic iir3 iir24 2>&1 | tail -3
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > gpurun_out/r06_iir_launches_call.txt 2>&1; tail -1 gpurun_out/r06_iir_launches_call.txt
( cd /tmp && rm -rf /tmp/iirt24 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt24 -- python $R/tools/exp/iir_stage_times.py 8 24 20 512 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt24 ) > gpurun_out/r06_iir_launches_call_bpo24.txt 2>&1; tail -1 gpurun_out/r06_iir_launches_call_bpo24.txt
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r06_bench_full.json 2>/dev/null | tail -1 > gpurun_out/r06_bench_image.json; wc -c gpurun_out/r06_bench_image.json
timeout 600 python bench.py --steps 50 --warmup 5 --layout packed --cpu-budget 0 --no-legs 2>&1 | tail -1 > gpurun_out/r06_bench_packed.json
