#!/bin/bash
# round 3, session k: kernel trace of the exact IIR bank (lane kernel + row scan), bpo 3 (C3 shape) and bpo 24
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
for cfg in "3 22 1024" "24 20 1024"; do
set -- $cfg
rm -rf $R/gpurun_out/prof/iir$1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $R/gpurun_out/prof/iir$1 -o t -- python $R/tools/bench_octbank.py --bpo $1 --log2-samples $2 --chunk $3 --iters 20 > $R/gpurun_out/prof/iir$1.log 2>&1 )
python tools/prof_summary.py stats gpurun_out/prof/iir$1/t_results.db | cut -c1-200 | sed -n 5,14p
python tools/exp/iir_stage_times.py gpurun_out/prof/iir$1 40 2>&1 | tail -42 | cut -c1-100
done
