#!/bin/bash
cat > /tmp/dp.py <<'PY'
import numpy as np, sys
sys.path.insert(0, "/root/repo")
from friture_amd.delay_estimator import DelayEstimatorStream
b = DelayEstimatorStream(1.0)
x = np.random.default_rng(0).standard_normal((2, 512))
for i in range(300):
    b.handle_new_data(x)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_delay -o d -- python /tmp/dp.py > /tmp/rp.log 2>&1
tail -3 /tmp/rp.log
find /tmp/prof_delay -type f | head
f=$(find /tmp/prof_delay -name '*kernel_stats.csv' | head -1); head -8 "$f"
f=$(find /tmp/prof_delay -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
f=$(find /tmp/prof_delay -name '*memory_copy_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("copies", len(d), "median us", d[len(d)//2] / 1e3)
PY
