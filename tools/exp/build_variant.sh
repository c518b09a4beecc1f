#!/bin/bash
# build_variant.sh <name> <source.hip> <extra flags...>: a copy of libfriture_hip.so with ONE translation unit recompiled
# with extra flags AND -DFRT_EXPERIMENTS (the FRT_* environment switches of tools/exp exist only in such builds), under
# tools/variants/<name>/ (use with LD_LIBRARY_PATH for A/B runs on the GPU box)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$2; shift 2
D=$R/tools/variants/$NAME
mkdir -p $D
EXTRA=""
case $SRC in
  stft.hip) EXTRA="-fno-slp-vectorize -Wno-inline-asm";;
  iir.hip) EXTRA="-ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form";;
  pipeline.hip|pitch.hip|specgram.hip) EXTRA="-ffp-contract=off";;
esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFRT_EXPERIMENTS -I$R/include $EXTRA "$@" -c $R/friture_amd/csrc/$SRC -o $D/${SRC%.hip}.o
OBJS=$(ls $R/friture_amd/lib/obj/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libfriture_hip.so $D/${SRC%.hip}.o $OBJS -Wl,-rpath,/opt/rocm/lib
echo built $D/libfriture_hip.so
