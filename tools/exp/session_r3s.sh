#!/bin/bash
# kernel times of the GCC paths at 100 pairs
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gcc -o g -- python /root/repo/tools/bench_gcc.py --pairs 100 --iters 10 > /tmp/rp.log 2>&1
tail -2 /tmp/rp.log
f=$(find /tmp/prof_gcc -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gcc" in r["Name"] or "any_" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us min {float(r["MinNs"])/1e3:8.1f}')
PY
