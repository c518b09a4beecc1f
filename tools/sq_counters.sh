#!/bin/bash
# SQ counters per kernel of any command: sq_counters.sh <outfile> <command...>   (six rocprofv3 --pmc passes, kernel-trace only)
# Per kernel (name, grid): dispatches, waves, instructions by class per wave, and the fractions of a wave's cycles (WAVE_CYCLES, in
# units of four clocks like the ACTIVE_* / WAIT_* counters) spent issuing VALU / anything, waiting for a counter (WAIT_INST_ANY) or for anything.
set -u
OUT=$1; shift
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/sqc
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_ADD_F64" \
            "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d /tmp/sqc/p$i -o p --output-format csv -- "$@" > /tmp/sqc_p$i.log 2>&1 || echo "pass $i failed"
done
python - > $OUT <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/sqc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:70], int(r["Grid_Size"]))
        acc[key][r["Counter_Name"][3:]].append(float(r["Counter_Value"]))
def avg(v, c): return sum(v[c]) / len(v[c]) if c in v and v[c] else 0.0
tot = lambda k: sum(acc[k].get("WAVE_CYCLES", [0]))
for key in sorted(acc, key=lambda k: -tot(k))[:40]:
    v = acc[key]; w = max(avg(v, "WAVES"), 1.0); wc = max(avg(v, "WAVE_CYCLES"), 1.0)
    print(f"{key[0]} grid {key[1]}: {len(v.get('WAVES', []))} dispatches, {w:.0f} waves, {4 * wc / w:.0f} clocks per wave")
    print("   per wave: " + "  ".join(f"{c.lower()} {avg(v, 'INSTS_' + c) / w:.0f}" for c in ("VALU", "SALU", "BRANCH", "SMEM", "LDS", "VMEM_RD", "VMEM_WR", "MFMA", "VALU_FMA_F64", "VALU_MUL_F64", "VALU_ADD_F64", "VALU_FMA_F32", "VALU_MUL_F32", "VALU_ADD_F32", "VALU_TRANS_F32", "VALU_CVT", "VALU_INT32", "VALU_INT64")))
    print("   of a wave's cycles: " + "  ".join(f"{n} {avg(v, c) / wc:.3f}" for n, c in (("VALU", "ACTIVE_INST_VALU"), ("scalar", "ACTIVE_INST_SCA"), ("LDS", "ACTIVE_INST_LDS"), ("VMEM", "ACTIVE_INST_VMEM"), ("any", "ACTIVE_INST_ANY"), ("wait-counter", "WAIT_INST_ANY"), ("wait-LDS", "WAIT_INST_LDS"), ("wait-any", "WAIT_ANY"))) + f"   LDS bank-conflict cycles / LDS active {avg(v, 'LDS_BANK_CONFLICT') / max(avg(v, 'LDS_IDX_ACTIVE'), 1):.3f}")
PY
