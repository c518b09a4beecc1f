#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r06_bench_full.json 2>/dev/null | tail -1 > gpurun_out/r06_bench_image.json; wc -c gpurun_out/r06_bench_image.json
timeout 600 python bench.py --steps 50 --warmup 5 --layout packed --cpu-budget 0 --no-legs 2>&1 | tail -1 > gpurun_out/r06_bench_packed.json
