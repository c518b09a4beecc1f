#!/bin/bash
# the round's closing check on a fresh box: whole GPU suite, smoke, the bench line + full record (profiles/r06_bench_*)
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 --full-json gpurun_out/r06_bench_full.json 2>/dev/null | tail -1 > gpurun_out/r06_bench_image.json; wc -c gpurun_out/r06_bench_image.json
timeout 600 python bench.py --steps 50 --warmup 5 --layout packed --cpu-budget 0 --no-legs 2>&1 | tail -1 > gpurun_out/r06_bench_packed.json
