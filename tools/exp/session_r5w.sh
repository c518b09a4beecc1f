#!/bin/bash
# round 5, session w: stft_pk16r_kernel with a frame's row stores issued behind the NEXT frame's first-stage loads (variant px; pxnd = stores
# at the end of their frame as before): parity of px, rates of both against the shipped kernel
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
B=tools/bin/stft_selftest
S="s/algorithmic.*of 8 TB.s)//; s/bench p32 N=16384 //"
echo "== parity (px)"
cp friture_amd/lib/libfriture_hip.so /tmp/base.so; cp tools/variants/px/libfriture_hip.so friture_amd/lib/libfriture_hip.so
FRT_STFT_PK16R=1 timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "large_frame or lds_staged or randomised" 2>&1 | tail -2
cp /tmp/base.so friture_amd/lib/libfriture_hip.so
FRT_LIB_VARIANT=px timeout 300 python tools/exp/pkr_debug.py 8192 2>&1 | grep -v amdgpu.ids | grep "repeatable\|max rel" 
echo "== rates, two rounds"
for rep in 1 2; do
  for cfg in "16384 8192 32 20 0" "16384 8192 32 20 3" "16384 4096 32 20 0" "16384 4096 32 20 3" "16384 8192 32 20 1"; do
    echo -n "shipped   : "; timeout 120 $B bench $cfg 0 40 | tail -1 | sed "$S"
    echo -n "pk16r px  : "; FRT_STFT_PK16R=1 LD_LIBRARY_PATH=$R/tools/variants/px timeout 120 $B bench $cfg 0 40 | tail -1 | sed "$S"
    echo -n "pk16r pxnd: "; FRT_STFT_PK16R=1 LD_LIBRARY_PATH=$R/tools/variants/pxnd timeout 120 $B bench $cfg 0 40 | tail -1 | sed "$S"
  done
done
