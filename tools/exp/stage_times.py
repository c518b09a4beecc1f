"""Per-stage durations of ola_batch_kernel from a rocprofv3 kernel trace: python tools/exp/stage_times.py <dir>"""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "ola_batch" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-9:]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
    print(f.split("/")[-2], " ".join(f"{x:7.1f}" for x in d), f"sum={sum(d):.1f}")
