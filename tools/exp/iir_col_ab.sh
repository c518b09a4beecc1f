#!/bin/bash
# column form of the lane pass: parity, then the same box's launch sequences with the form off / by the shipped rule / everywhere, and 64-double tiles
set -u
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_iir_gpu.py -x -q -m gpu 2>&1 | tail -2
for cfg in "8 3 22 1024" "8 24 20 512"; do
  echo "#### $cfg"
  echo "-- column form off";        FRT_OPTIONS="iir_lane_columns=0" bash tools/exp/iir_trace_variants.sh "$cfg" c32 | grep -v "split\|^=="
  echo "-- shipped rule";           bash tools/exp/iir_trace_variants.sh "$cfg" c32 | grep -v "split\|^=="
  echo "-- everywhere";             FRT_OPTIONS="iir_lane_columns=1" bash tools/exp/iir_trace_variants.sh "$cfg" c32 | grep -v "split\|^=="
  echo "-- shipped rule, 64-double tiles"; bash tools/exp/iir_trace_variants.sh "$cfg" c64 | grep -v "split\|^=="
done
