#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -3
for ch in 4096 2048; do python tools/bench_octbank.py --chunk $ch 2>/dev/null | tail -1 | cut -c1-200; done
python tools/bench_octbank.py --chunk 4096 --channels 64 --bpo 24 --log2-samples 20 2>/dev/null | tail -1 | cut -c1-200
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/oct2048 -o oct -- python $GRAFT_REPO_ROOT/tools/bench_octbank.py --chunk 2048 --iters 3 > /dev/null 2>&1 )
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("gpurun_out/prof/oct2048/*.db")[0])
for r in db.execute("select name, count(*), avg(duration)/1e3, sum(duration)/1e3 from kernels group by name order by name"):
    print("  %-40s calls %3d avg %9.1f us total %9.1f us" % (r[0][:40], r[1], r[2], r[3]))
PY
