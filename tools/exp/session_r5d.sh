#!/bin/bash
# round 5, session d: (1) the exact bank's output pass with only its band groups / only its decimator launched (variant ix, FRT_LANE_ONLY):
# does a band wave cost what lane_probe says, or twice that?  (2) non-temporal stores on split rows, aligned sets
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
for mode in both b d; do
  OUT=$R/gpurun_out/iir_lane_$mode; rm -rf $OUT; mkdir -p $OUT
  if [ $mode = both ]; then unset FRT_LANE_ONLY; else export FRT_LANE_ONLY=$mode; fi
  ( cd /tmp && FRT_LIB_VARIANT=ix timeout 200 rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python $R/tools/exp/iir_stage_times.py 8 3 22 > $OUT.log 2>&1 )
  echo "== lane groups: $mode"; python tools/exp/iir_stage_times.py --parse $OUT | grep -E "lane|launches"
done
unset FRT_LANE_ONLY
B=tools/bin/stft_selftest
S="s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//"
for rep in 1 2; do
for v in base nt ntl; do
  if [ $v = base ]; then LP=""; else LP=$R/tools/variants/$v; fi
  for cfg in "3 0 40 32 0" "3 0 40 32 1" "0 0 40 32 1"; do
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
  done
done
done
