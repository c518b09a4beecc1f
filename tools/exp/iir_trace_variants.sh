#!/bin/bash
# launch sequence of one IirBank.energies call per variant library: iir_trace_variants.sh "<ch bpo log2n chunk>" <variants...> ("intree" = the in-tree library)
set -u
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
CFG=$1; shift
for v in "$@"; do
  D=/tmp/iirt_$v; rm -rf $D
  if [ "$v" = intree ]; then env -u FRT_LIB_VARIANT timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/tools/exp/iir_stage_times.py $CFG > /dev/null 2>&1
  else FRT_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/tools/exp/iir_stage_times.py $CFG > /dev/null 2>&1; fi
  echo "== $v ($CFG)"; python $R/tools/exp/iir_stage_times.py --parse $D | grep "${IIR_TRACE_FILTER:-lane\|launches}"
done
