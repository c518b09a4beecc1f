#!/bin/bash
# round 6, session a: the resident GCC-PHAT kernel (parity, rates at 512 / 768 threads, phase stamps, PMC traffic of both kernels),
# where the headline's fixed host time goes.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== gcc tests"; timeout 900 python -m pytest tests/test_gcc_gpu.py -x -q 2>&1 | tail -5
echo "== gcc batch rates (shipped library: resident kernel at 512 threads)"
timeout 600 python tools/bench_gcc.py --pairs 1 32 64 100 160 256 512 1024 2>&1 | grep -v "^{" | tee $O/gcc_batch_512.txt
echo "== variant: resident kernel at 768 threads"
FRT_LIB_VARIANT=res768 timeout 600 python tools/exp/gcc_variant_bench.py --pairs 100 256 1024 2>&1 | grep -v "^{" | tee $O/gcc_batch_768.txt
echo "== phase stamps (experiments build, 512 threads), 1 pair / 256 / 1024"
for p in 1 256 1024; do FRT_GCC_PROFILE=1 FRT_LIB_VARIANT=res512x timeout 300 python tools/exp/gcc_variant_bench.py --pairs $p --iters 2 2>&1 | grep "resident_kernel phases" | tail -2; done | tee $O/gcc_phases.txt
echo "== PMC traffic: gcc legs (resident kernel)"
bash tools/gpu_leg_traffic.sh r6a/traffic gcc1024 gcc100 2>&1 | tail -8
echo "== PMC traffic: the kernel with the scratch slab, 1024 pairs"
LEG_TRAFFIC_ARGS="--set-option gcc_resident=0" LEG_TRAFFIC_JSON=$O/traffic_slab.json bash tools/gpu_leg_traffic.sh r6a/traffic_slab gcc1024 2>&1 | tail -2
echo "== bench headline only: wall against events on this box"
timeout 600 python bench.py --steps 20 --warmup 5 --no-legs --cpu-budget 0 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['kernel_ms_repeats'], r['value'])"
