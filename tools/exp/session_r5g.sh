#!/bin/bash
# round 5, session g: from which stage on should the overlap-add bank run its decimators ahead?  (variant ox, FRT_OLA_DEFER_BELOW = sets x channels)
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for thr in 0 128 400 1024 2048 4096; do
  echo -n "defer below $thr: "; FRT_OLA_DEFER_BELOW=$thr FRT_LIB_VARIANT=ox timeout 200 python tools/bench_firbank.py 2>/dev/null | python -c "
import json,sys
print(' | '.join('%d ch bpo %d: %.4f ms' % (d['channels'], d['bpo'], d['fir']['ms']) for d in map(json.loads, sys.stdin)))"
done
done
