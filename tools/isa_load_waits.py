#!/usr/bin/env python
"""Loads that are awaited right where they are issued, inside loops — the pattern that hid in the exact bank's output passes from round 3
to round 6 (a prefetch whose merge copies the compiler placed behind the loads: global_load x4, s_waitcnt vmcnt(3..0), v_mov x16).

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S x.hip -o x.s ;  python tools/isa_load_waits.py x.s [kernel-name-substring]

For every loop (label .. backward branch) of every kernel: the vector-memory loads issued in the loop and, for each, how many
instructions (and how many float64 / MFMA / LDS instructions) lie between it and the first s_waitcnt vmcnt(n) that covers it on the
fall-through path (loads and stores retire in order on gfx9: a wait for vmcnt(n) covers everything but the n youngest).  Reported: loops in
which a load is covered within MAXGAP instructions.  Control flow inside the loop is ignored (straight-line approximation)."""
import re
import sys

MAXGAP = 6


def kernels(text):
    cur, name = None, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def is_vm(op):
    return op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store", "scratch_", "global_atomic", "buffer_atomic"))


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, lines in kernels(text):
        if want not in name:
            continue
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(lines):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        # innermost loops only
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        for a, b in inner:
            body = [x.strip() for x in lines[a:b + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
            ops = [x.split()[0] for x in body]
            vm_idx = [i for i, o in enumerate(ops) if is_vm(o)]
            hits = []
            for k, i in enumerate(vm_idx):
                if "load" not in ops[i]:
                    continue
                issued = k + 1                                       # vm ops issued up to and including this load
                for j in range(i + 1, len(ops)):
                    if is_vm(ops[j]):
                        issued += 1
                    if ops[j] == "s_waitcnt":
                        m = re.search(r"vmcnt\((\d+)\)", body[j])
                        if m and issued - int(m.group(1)) >= k + 1:
                            gap = j - i - 1
                            if gap <= MAXGAP:
                                hits.append((i, gap, body[i][:60]))
                            break
            if hits:
                work = sum(1 for o in ops if "f64" in o or "mfma" in o or o.startswith("v_pk_") or o.startswith("v_fma"))
                print(f"{name[:90]}\n   loop lines {a}-{b}: {len(ops)} instructions ({work} fma/mfma), {len(hits)} load(s) awaited within {MAXGAP} instructions:")
                for i, gap, txt in hits[:4]:
                    print(f"      +{i:4d} gap {gap}: {txt}")


if __name__ == "__main__":
    main()
