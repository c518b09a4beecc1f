#!/bin/bash
# refresh of the bench-derived profile files (K1 sources unchanged: the PMC traffic file stays valid)
R=$GRAFT_REPO_ROOT; TAG=r02; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
timeout 900 python bench.py --steps 50 --warmup 5 2>gpurun_out/bench_image.err | tail -1 > gpurun_out/${TAG}_bench_image.json
timeout 600 python bench.py --steps 50 --warmup 5 --kind psd --cpu-budget 0 --no-legs 2>&1 | tail -1 > gpurun_out/${TAG}_bench_psd.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --cpu-budget 0 > $R/gpurun_out/prof/stats.log 2>&1 )
python tools/prof_summary.py stats gpurun_out/prof/stats/bench_results.db > gpurun_out/${TAG}_bench_kernel_stats.txt
python tools/bench_firbank.py > gpurun_out/${TAG}_banks.json 2>/dev/null
cut -c1-200 gpurun_out/${TAG}_banks.json
