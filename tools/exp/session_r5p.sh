#!/bin/bash
# round 5, session p: stft_pk16r_kernel (N = 16384 with two workgroups per CU: register ring, streamed constants) — (1) parity of the
# large-frame tests on the shipped library, (2) rates against stft_pk16_kernel (variant px with FRT_STFT_NO_PK16R=1) and between the
# places a frame requests the next one's samples (variants px / p0 / p3 / p2, see build lines in tools/exp/README.md), (3) run lengths
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
echo "== (1) parity"
timeout 300 $B check | tail -2
timeout 900 python -m pytest tests/test_stft_gpu.py -x -q -k "large_frame or lds_staged or randomised or generic_hops or all_sizes" 2>&1 | tail -4
S="s/algorithmic.*of 8 TB.s)//"
bench() { # variant, label, env assignment or "-", args
  local v=$1 label=$2 envs=$3; shift 3
  echo -n "$label: "; ( [ "$envs" != "-" ] && export $envs; LD_LIBRARY_PATH=$R/tools/variants/$v:${LD_LIBRARY_PATH:-} timeout 120 $B bench "$@" | tail -1 | sed "$S" )
}
echo "== (2) A/B, two rounds"
for rep in 1 2; do
  for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3" "16384 8192 32 20 1"; do
    bench px "pk16 (1 WG/CU)  " FRT_STFT_NO_PK16R=1 $cfg 0 40
    bench px "pk16r px        " - $cfg 0 40
    bench p0 "pk16r p0        " - $cfg 0 40
    bench p3 "pk16r p3        " - $cfg 0 40
    bench p2 "pk16r p2        " - $cfg 0 40
  done
done
echo "== (3) run lengths (px)"
for run in 4 6 8 12 16 32; do
  for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0"; do
    bench px "pk16r px run=$run" - $cfg $run 40
  done
done
echo "== (4) shipped library"
for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3"; do
  echo -n "shipped: "; timeout 120 $B bench $cfg 0 40 | tail -1 | sed "$S"
done
