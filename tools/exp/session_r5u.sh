#!/bin/bash
# round 5, session u: GCC-PHAT, the default window in small batches as FOUR sub-transforms of 3000 points (compile-time plan 3 x 10 x 10 x 10,
# 48 KB of LDS: three workgroups per CU, eight forward + four inverse workgroups per pair) instead of two of 6000: parity, rates by batch size
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
echo "== parity"; timeout 600 python -m pytest tests/test_gcc_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -3
echo "== rates by batch size (windows/s; default = the shape rule, one_workgroup / split forced)"
python tools/bench_gcc.py 2>/dev/null | grep -v "^{"
echo "== launches of one call, 100 pairs and 1 pair"
for n in 100 1; do
( cd /tmp && rm -rf /tmp/gcct && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gcct -- python $R/tools/bench_gcc.py --pairs $n > /dev/null 2>&1
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("/tmp/gcct/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])))
rows.sort()
last=[i for i,r in enumerate(rows) if "argmax_combine" in r[2]]
if last:
    b=last[-1]+1; a=b-5
    t0=rows[a][0]
    for s,e,n,gx,gy in rows[a:b]: print(f"{(s-t0)/1e3:8.1f} us  {n:40s} grid {gx:5d} x {gy:4d}  {(e-s)/1e3:7.1f} us")
    print(f"span {(rows[b-1][1]-t0)/1e3:.1f} us")
PY
)
done
