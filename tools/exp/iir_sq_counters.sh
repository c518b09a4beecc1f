#!/bin/bash
# SQ counters of the exact bank's launches (one IirBank.energies call x 6): iir_sq_counters.sh "<ch bpo log2n chunk>" [variant]
set -u
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
CFG=${1:-8 3 22 1024}; V=${2:-}
[ -n "$V" ] && export FRT_LIB_VARIANT=$V
rm -rf /tmp/sq
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
            "SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/sq/p$i -o p --output-format csv -- python $R/tools/exp/iir_stage_times.py $CFG > /tmp/sq_p$i.log 2>&1 || echo "pass $i failed"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0][-34:], int(r["Grid_Size"]) if "Grid_Size" in r else 0)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
for key in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0]))):
    if "lane" not in key[0] and "zero_state" not in key[0]: continue
    v = acc[key]
    print(f"{key[0]} grid {key[1]}  ({len(v.get('SQ_WAVES', []))} dispatches)")
    print("   " + "  ".join(f"{c[3:]}={sum(v[c]) / len(v[c]):.4g}" for c in names if c in v))
PY
