#!/bin/bash
# kernel-level timeline of the streaming objects' chunk pushes (last pushes of a run): where a chunk's device time goes
cat > /tmp/ts.py <<'PY'
import numpy as np, sys
from fractions import Fraction
sys.path.insert(0, "/root/repo")
which = sys.argv[1]
rng = np.random.default_rng(0)
x = 0.25 * rng.standard_normal((1, 512))
if which == "octave":
    from friture_amd.octavespectrum import OctaveSpectrumStream
    o = OctaveSpectrumStream(3)
    f = o.handle_new_data
elif which == "octave_bb":
    from friture_amd.octavespectrum import OctaveSpectrum
    o = OctaveSpectrum(3)
    f = o.handle_new_data
elif which == "specgram":
    from friture_amd.spectrogram import SpectrogramStream
    o = SpectrogramStream(fft_size=1024, overlap=Fraction(3, 4), weighting=1, screen_width=800, screen_height=400, timerange_s=10.0)
    f = o.handle_new_data
elif which == "specgram_bb":
    from friture_amd.spectrogram import Spectrogram
    o = Spectrogram(fft_size=1024, overlap=Fraction(3, 4), weighting=1, screen_width=800, screen_height=400, timerange_s=10.0)
    f = o.handle_new_data
import time
for i in range(200):
    f(x)
    time.sleep(0.0005)
PY
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rm -rf /tmp/prof_$w
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_$w -o d -- python /tmp/ts.py $w > /tmp/rp_$w.log 2>&1
  echo "=== $w"
  python - /tmp/prof_$w <<'PY'
import csv, sys, glob
d = sys.argv[1]
rows = []
for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]))
rows.sort()
rows = rows[-36:]
t0 = rows[0][0]
for s, e, n in rows:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {n}")
PY
done
