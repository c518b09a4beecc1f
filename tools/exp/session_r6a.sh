#!/bin/bash
# round 6, session a: the resident GCC-PHAT kernel (parity, rates at 512 / 768 threads, phase stamps, PMC traffic of both kernels),
# where the headline's fixed host time goes.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== gcc tests"; timeout 900 python -m pytest tests/test_gcc_gpu.py -x -q 2>&1 | tail -5
echo "== gcc batch rates (shipped library: resident kernel at 768 threads)"
timeout 600 python tools/bench_gcc.py --pairs 1 32 64 100 160 256 512 1024 2>&1 | grep -v "^{" | tee $O/gcc_batch_768.txt
echo "== variant: resident kernel at 512 threads"
FRT_LIB_VARIANT=res512 timeout 600 python tools/exp/gcc_variant_bench.py --pairs 100 256 1024 2>&1 | grep -v "^{" | tee $O/gcc_batch_512.txt
echo "== phase stamps (experiments build, 768 threads), 1 pair / 256 / 1024"
for p in 1 256 1024; do FRT_GCC_PROFILE=1 FRT_LIB_VARIANT=res768x timeout 300 python tools/exp/gcc_variant_bench.py --pairs $p --iters 2 2>&1 | grep "phases" | tail -2; done | tee $O/gcc_phases.txt
echo "== host fixed time of the headline's timed region"
timeout 300 python tools/exp/host_fixed.py | tee $O/host_fixed.json
timeout 300 python tools/exp/host_fixed.py --spin | tee $O/host_fixed_spin.json
echo "== PMC traffic: gcc legs (resident kernel)"
bash tools/gpu_leg_traffic.sh r6a/traffic gcc1024 gcc100 2>&1 | tail -8
