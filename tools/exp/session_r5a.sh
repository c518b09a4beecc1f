#!/bin/bash
# round 5, session a: the split output rows (frt_stft_run_split) — parity first, then packed vs split rows of the headline
# kernel with plain / non-temporal / LDS-staged 16-byte stores (tools/variants/{nt,ntl,x4,x4nt,x4ntl}), run lengths auto (16) / 8 / 4
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
echo "== parity of the split layout (shipped library)"
timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "split" 2>&1 | tail -5
B=tools/bin/stft_selftest
for v in base nt ntl x4 x4nt x4ntl; do
  if [ $v = base ]; then LP=""; else LP=$R/tools/variants/$v; fi
  for cfg in "3 0 40 32 0" "3 0 40 32 1" "3 8 40 32 1" "3 4 40 32 1" "0 0 40 32 0" "0 0 40 32 1" "0 8 40 32 1"; do
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "s/algorithmic.*of 8 TB.s)//"
  done
done
echo "== second pass of the headline lines (box drift)"
for v in base x4 x4nt x4ntl nt; do
  if [ $v = base ]; then LP=""; else LP=$R/tools/variants/$v; fi
  for cfg in "3 0 40 32 0" "3 0 40 32 1"; do
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "s/algorithmic.*of 8 TB.s)//"
  done
done
echo "== parity of the x4 variants (library swapped in the scratch copy)"
cp friture_amd/lib/libfriture_hip.so /tmp/base.so
for v in x4ntl; do
  cp tools/variants/$v/libfriture_hip.so friture_amd/lib/libfriture_hip.so
  timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "split" 2>&1 | tail -3
done
cp /tmp/base.so friture_amd/lib/libfriture_hip.so
echo "== the float64 instance, packed vs split"
for cfg in "0 0 30 64 0" "0 0 30 64 1" "3 0 30 64 0" "3 0 30 64 1"; do
  echo -n "base: "; timeout 120 $B bench 1024 512 1 25 $cfg | tail -1 | sed "s/algorithmic.*of 8 TB.s)//"
done
