#!/bin/bash
# round 5, session v: SQ counters of the N = 16384 PSD kernels (32 ch x 2^20, hop N/2): the shipped stft_pk16_kernel, stft_pk16r_kernel
# (variant px, FRT_STFT_PK16R=1) and its build without any global access (variant a27) — how busy are the vector pipe and the LDS?
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
run() { # tag, variant dir or "", env
  local tag=$1 lib=$2 envs=$3
  ( [ "$envs" != "-" ] && export $envs; [ -n "$lib" ] && export LD_LIBRARY_PATH=$R/tools/variants/$lib:${LD_LIBRARY_PATH:-}; bash tools/gpu_pmc.sh $tag 0 0 16384 8192 32 20 sq > /dev/null 2>&1 )
  python tools/prof_summary.py pmc gpurun_out/pmc_$tag stft_pk | grep -v "^#" | sed "s/^/$tag  /"
}
run r5v_pk16 "" -
run r5v_pk16r px FRT_STFT_PK16R=1
run r5v_pk16r_a27 a27 FRT_STFT_PK16R=1
