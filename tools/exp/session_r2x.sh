#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_iir; rm -rf $OUT; mkdir -p $OUT
CMD="python $R/tools/bench_octbank.py --iters 2"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS -d $OUT/sq1 -o p --output-format csv -- $CMD > /dev/null 2>&1
cd $R
python - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_iir/sq1/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
# group by dispatch id for iir_stage_kernel, take the biggest dispatches (stage 0 output pass)
by = collections.defaultdict(dict)
for r in rows:
    if 'iir_stage_kernel' in r['Kernel_Name']:
        by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
big = sorted(by.values(), key=lambda d: -d.get('SQ_WAVES', 0))[:3]
for d in big:
    w = d['SQ_WAVES']
    print({k: (round(v / w, 1) if k != 'SQ_WAVES' else v) for k, v in d.items()}, 'valu_active/wave_cycles=%.3f' % (d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES']))
P
