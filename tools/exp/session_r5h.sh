#!/bin/bash
# round 5, session h: overlap-add bank, decimators ahead: threshold (sets x channels) against bands per octave
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
for cfg in "8 1 22" "8 6 21" "8 12 20" "8 24 20" "64 24 20"; do
for thr in 0 4096 16384 1000000000; do
  echo -n "ch bpo log2n = $cfg, defer below $thr: "; FRT_OLA_DEFER_BELOW=$thr FRT_LIB_VARIANT=ox timeout 200 python tools/exp/fir_only.py $cfg 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms' % d['ms'], d['digest'])"
done
done
