"""Per-launch durations of the IIR bank's kernels (last batch) from a rocprofv3 kernel trace: python tools/exp/iir_stage_times.py <dir> [launches per batch]"""
import csv, glob, sys
per = int(sys.argv[2]) if len(sys.argv) > 2 else 37
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "iir_" in r["Kernel_Name"] or "energy_scan" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-per:]
    t0 = int(last[0]["Start_Timestamp"])
    tot = {}
    for r in last:
        name = r["Kernel_Name"].split("(")[0].replace("frt::", "")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[name] = tot.get(name, 0) + d
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {name}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
    print("totals:", {k: round(v, 1) for k, v in tot.items()}, "span", (int(last[-1]["End_Timestamp"]) - t0) / 1e3)
