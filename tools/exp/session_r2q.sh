#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_iir_gpu.py -x -q -m gpu 2>&1 | tail -5
python tools/bench_octbank.py --iters 10 | cut -c1-80
FRT_ZS_VECTOR=1 python tools/bench_octbank.py --iters 10 | cut -c1-80
cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/iir_trace/base -o p --output-format csv -- python $R/tools/bench_octbank.py --iters 3 > /dev/null 2>&1
cd $R; python tools/exp/iir_stage_times.py gpurun_out/iir_trace 40 | grep "zero_state\|totals" | awk '{print $1,$2,$3}' | tr '\n' ';'
