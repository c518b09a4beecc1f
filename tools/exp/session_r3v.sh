#!/bin/bash
# R = 4 (3000-point sub-transforms, run-time plan, 1024 threads) forced for L = 24000: what the deeper split costs per launch
cd /tmp && export TMPDIR=/tmp
FRT_GCC_FORCE_R=4 FRT_GCC_ONE_WORKGROUP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gcc4 -o g -- python /root/repo/tools/bench_gcc.py --pairs 100 --iters 10 > /tmp/rp.log 2>&1
grep "^100" /tmp/rp.log
f=$(find /tmp/prof_gcc4 -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gcc" in r["Name"] or "any_" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us min {float(r["MinNs"])/1e3:8.1f}')
PY
