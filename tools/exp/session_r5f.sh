#!/bin/bash
# round 5, session f: FFT overlap-add bank with the low-rate stages' decimators ahead and their band filters in ONE launch
# (ola_pair_multi_kernel) — parity, then A/B against the per-stage launches (variant ox, FRT_OLA_NO_DEFER=1), then the launch list
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
echo "== GPU suite"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
  echo "== deferred (variant ox, default)"; FRT_LIB_VARIANT=ox timeout 200 python tools/bench_firbank.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['channels'],d['bpo'],'fir %.4f ms  iir %.4f ms' % (d['fir']['ms'], d['iir']['ms']), d['fir']['digest'])"
  echo "== per-stage launches (FRT_OLA_NO_DEFER=1)"; FRT_OLA_NO_DEFER=1 FRT_LIB_VARIANT=ox timeout 200 python tools/bench_firbank.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['channels'],d['bpo'],'fir %.4f ms  iir %.4f ms' % (d['fir']['ms'], d['iir']['ms']), d['fir']['digest'])"
done
OUT=$R/gpurun_out/ola_trace; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python $R/tools/exp/fir_only.py > $OUT.log 2>&1 )
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/ola_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-30:], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
rows.sort()
names=[r[2] for r in rows]
ends=[i for i,n in enumerate(names) if "energy_finish" in n or "energy_scan" in n]
if len(ends)>=2:
    a,b=ends[-2]+1,ends[-1]+1
    t0=rows[a][0]
    for s,e,n,gx,gy,gz in rows[a:b]: print(f"{(s-t0)/1e3:8.1f} us  {n:32s} grid {gx:6d} x {gy:3d} x {gz:3d}  {(e-s)/1e3:7.1f} us")
    print(f"{b-a} launches, span {(rows[b-1][1]-t0)/1e3:.1f} us")
PY
