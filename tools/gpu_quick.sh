#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -4
for ch in 16384 8192 4096; do python tools/bench_octbank.py --chunk $ch 2>/dev/null | tail -1 | cut -c1-200; done
export TMPDIR=/tmp
for ch in 16384; do
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/oct$ch -o oct -- python $GRAFT_REPO_ROOT/tools/bench_octbank.py --chunk $ch --iters 3 > /dev/null 2>&1 )
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("gpurun_out/prof/oct$ch/*.db")[0])
print("chunk $ch")
for r in db.execute("select name, grid_x, count(*), avg(duration)/1e3, sum(duration)/1e3 from kernels group by name, grid_x order by name, grid_x desc"):
    print("  %-40s grid %8d calls %3d avg %9.1f us total %9.1f us" % (r[0][:40], r[1], r[2], r[3], r[4]))
rows=list(db.execute("select name, duration/1e3 from kernels where name like '%iir_stage%' order by start"))
print("stage launches of the last call:", [round(d) for n,d in rows[-18:]])
PY
done
