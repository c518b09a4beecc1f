"""Condense rocprofv3 output (rocpd .db from --kernel-trace --stats, CSV from --pmc passes) into
small text summaries that are committed under profiles/.

    python tools/prof_summary.py stats <results.db> > profiles/rNN_kernel_stats.txt
    python tools/prof_summary.py pmc <dir-with-pass-subdirs> <kernel-substring> > profiles/rNN_pmc.txt
"""
import collections
import csv
import glob
import os
import sqlite3
import sys


def stats(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    print(f"# rocprofv3 --kernel-trace --stats  ({os.path.basename(db_path)})")
    print("# vgpr / agpr: registers allocated per lane = 2 x the trace's VGPR_Count / Accum_VGPR_Count fields (on gfx950 the trace")
    print("#   reports the allocation in units of two registers: 76 for the 149 -> 152 registers of stft_kernel<9,4>; cross-checked")
    print("#   against hipcc -Rpass-analysis=kernel-resource-usage); waves/SIMD = min(8, 512 // (vgpr + agpr))")
    print("# name | calls | total_us | avg_us | min_us | max_us | % | vgpr | agpr | waves/SIMD | sgpr | lds_bytes | grid (threads x,y,z) | workgroup (x,y,z)")
    total = cur.execute("select sum(duration) from kernels").fetchone()[0]
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), vgpr_count, accum_vgpr_count, sgpr_count, "
         "lds_size, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z from kernels group by name order by sum(duration) desc")
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, gx, gy, gz, wx, wy, wz in cur.execute(q):
        grid, wg = f"{gx}x{gy}x{gz}", f"{wx}x{wy}x{wz}"          # the full launch shape: a 4 x 100 grid is not 4 workgroups
        short = name if len(name) < 100 else name[:97] + "..."
        vg, ag = 2 * (vg or 0), 2 * (ag or 0)
        occ = min(8, 512 // max(vg + ag, 1))
        print(f"{short} | {n} | {tot/1e3:.1f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.1f} | "
              f"{vg} | {ag} | {occ} | {sg} | {lds} | {grid} | {wg}")


def pmc(root, needle):
    print(f"# rocprofv3 --pmc passes under {root}; kernels matching '{needle}'; mean over dispatches")
    for d in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(d)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{os.path.basename(os.path.dirname(d)):8s} {k:28s} dispatches={len(v):3d} mean={sum(v)/len(v):.6g} sum={sum(v):.6g} max={max(v):.6g}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
