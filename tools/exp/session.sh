#!/bin/bash
# One parametrised GPU session (replaces the 91 one-off session_r2*.sh … session_r5*.sh of rounds 2-5; those are in git history up to
# commit 196789f).  Parity first, then the same bench tool against the in-tree library and each variant, optionally a kernel trace.
#
#   gpurun --timeout 1800 -- 'bash tools/exp/session.sh <tag> "<pytest args>" "<bench command with {V}>" "<variants>" [trace]'
#
#   <tag>            outputs go to gpurun_out/<tag>/
#   <pytest args>    e.g. "tests/test_gcc_gpu.py -k resident"   ("" = skip)
#   <bench command>  run once per variant with FRT_LIB_VARIANT set ({V} in the command is replaced by the variant's name, "base" for
#                    the in-tree library), e.g. "python tools/exp/gcc_variant_bench.py --pairs 100 1024"
#   <variants>       names under tools/variants/ (tools/exp/build_variant.sh), e.g. "res768 look0"; the in-tree library always runs first
#   trace            also run the bench command of the in-tree library under rocprofv3 --kernel-trace --stats
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:?tag}; TESTS=${2:-}; BENCH=${3:-}; VARIANTS=${4:-}; TRACE=${5:-}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ -n "$TESTS" ]; then echo "== pytest $TESTS"; timeout 1500 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt; fi
if [ -n "$BENCH" ]; then
  for v in base $VARIANTS; do
    echo "== $v"
    cmd=${BENCH//\{V\}/$v}
    if [ "$v" = base ]; then env -u FRT_LIB_VARIANT timeout 600 $cmd 2>&1 | grep -v amdgpu.ids | tee $O/bench_$v.txt
    else FRT_LIB_VARIANT=$v timeout 600 $cmd 2>&1 | grep -v amdgpu.ids | tee $O/bench_$v.txt; fi
  done
  if [ -n "$TRACE" ]; then
    ( cd /tmp && rm -rf /tmp/sess_trace && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sess_trace -o t -- ${BENCH//\{V\}/base} > /dev/null 2>&1 )
    python tools/prof_summary.py stats /tmp/sess_trace/t_results.db > $O/kernel_stats.txt 2>/dev/null; head -12 $O/kernel_stats.txt | cut -c1-180
  fi
fi
