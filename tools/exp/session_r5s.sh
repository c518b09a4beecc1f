#!/bin/bash
# round 5, session s: (1) the whole GPU suite on the library as it now ships (device code compressed, N = 8192 / 4096 / 2048 from one
# template, no register-window instances of the generic kernel for N >= 2048); (2) stft_pk16r_kernel with the store-data pad
# (variant px, FRT_STFT_PK16R=1): large-frame tests and the bin-by-bin comparison with the shipped kernel on the full-size shard
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
echo "== (1) selftest + GPU suite"; timeout 300 tools/bin/stft_selftest check | tail -1
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | grep -v "^$\|user\|sys"
echo "== (2) stft_pk16r_kernel with the pad"
cp friture_amd/lib/libfriture_hip.so /tmp/base.so
cp tools/variants/px/libfriture_hip.so friture_amd/lib/libfriture_hip.so
FRT_STFT_PK16R=1 timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "large_frame or lds_staged or randomised" 2>&1 | tail -3
cp /tmp/base.so friture_amd/lib/libfriture_hip.so
FRT_LIB_VARIANT=px timeout 300 python tools/exp/pkr_debug.py 8192 2>&1 | grep -v amdgpu.ids | tail -12
