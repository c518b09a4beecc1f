#!/bin/bash
# round 6, session f: where a chunk-scan launch's time goes (ablation builds: no arithmetic in the walks / no end-state loads / neither)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in "" scanA1 scanA2 scanA3; do
  echo "== variant '${v:-shipped}'"
  ( cd /tmp && rm -rf /tmp/iirt && FRT_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) 2>&1 | grep "scan_kernel\|launches" | tee $O/scan_${v:-shipped}.txt
done
