#!/bin/bash
# PMC traffic of the bench legs' kernels.  usage: gpu_leg_traffic.sh <outdir under gpurun_out> [legs ...]
# One rocprofv3 run per counter set (kernel-trace only beside --pmc), 4 identical calls of the leg's workload each.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-leg_traffic}; shift
LEGS=${@:-$(python $R/tools/leg_traffic.py legs)}
export TMPDIR=/tmp
cd /tmp
for leg in $LEGS; do
  mkdir -p $OUT/$leg
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    n=$(echo $pass | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $OUT/$leg/$n -o p --output-format csv -- python $R/tools/leg_traffic.py run $leg --calls 4 --out $OUT/$leg ${LEG_TRAFFIC_ARGS:-} > $OUT/$leg/$n.log 2>&1
    echo "leg $leg pass $n rc=$?"
  done
done
cd $R && python tools/leg_traffic.py collect $OUT ${LEG_TRAFFIC_JSON:-} && cp ${LEG_TRAFFIC_JSON:-profiles/r06_leg_traffic.json} $OUT/
