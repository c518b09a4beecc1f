#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -4
for ch in 4096 2048 1024 -2048; do python tools/bench_octbank.py --chunk $ch 2>/dev/null | tail -1 | cut -c1-200; done
python tools/bench_octbank.py --chunk 4096 --channels 64 --bpo 24 --log2-samples 20 2>/dev/null | tail -1 | cut -c1-200
python tools/bench_octbank.py --chunk 2048 --channels 64 --bpo 24 --log2-samples 20 2>/dev/null | tail -1 | cut -c1-200
export TMPDIR=/tmp
for ch in 2048; do
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/oct$ch -o oct -- python $GRAFT_REPO_ROOT/tools/bench_octbank.py --chunk $ch --iters 3 > /dev/null 2>&1 )
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("gpurun_out/prof/oct$ch/*.db")[0])
print("chunk $ch")
for r in db.execute("select name, count(*), avg(duration)/1e3, sum(duration)/1e3 from kernels group by name order by name"):
    print("  %-40s calls %3d avg %9.1f us total %9.1f us" % (r[0][:40], r[1], r[2], r[3]))
rows=list(db.execute("select name, duration/1e3, grid_x from kernels where name like '%iir_%' order by start"))
print("launches of the last call:", [(n[9:14], round(d)) for n,d,g in rows[-36:]])
PY
done
