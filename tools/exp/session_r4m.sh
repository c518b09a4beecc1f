#!/bin/bash
# round 4, session m: what the IIR bank's kernels wait for (SQ counters over the configs[2] batch)
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_r4m
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/tools/bench_octbank.py --bpo 3 --log2-samples 22 --channels 8 --chunk 1024 --iters 3"
pass() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- $CMD > $OUT/$name.log 2>&1 ); echo "pass $name rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F64
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_r4m/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "iir_lane" in k or "zero_state" in k or "iir_scan" in k:
        # the largest dispatches of each kernel (stage 0)
        wc=v.get("SQ_WAVE_CYCLES",[0]); m=max(wc)
        idx=[i for i,x in enumerate(wc) if x>0.8*m]
        def big(name):
            a=v.get(name,[])
            return sum(a[i] for i in idx if i<len(a))/max(1,len(idx)) if a else float('nan')
        print(k, "largest dispatches:", len(idx))
        for name in sorted(v): 
            a=v[name]; mm=max(a); ii=[i for i,x in enumerate(a) if x>0.8*mm]
            print("   %-26s %.4g"%(name, sum(a[i] for i in ii)/len(ii)))
PY
