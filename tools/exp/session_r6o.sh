#!/bin/bash
# round 6, session o: the chunk scan's rows of 4th-order filters as quads — parity, bank times, launch lists (27 and 216 bands)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6o; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== iir tests"; timeout 1500 python -m pytest tests/test_iir_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -4
for cfg in "--bpo 3 --log2-samples 22 --chunk 1024" "--bpo 24 --log2-samples 20 --chunk 512" "--bpo 24 --log2-samples 20 --chunk 1024" "--bpo 12 --log2-samples 20 --chunk 512" "--channels 64 --bpo 24 --log2-samples 20 --chunk 512"; do
  timeout 300 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['channels'], r['bpo'], r['chunk'], round(r['ms'],4), 'ms')"
done 2>&1 | tee $O/bank_times.txt
( cd /tmp && rm -rf /tmp/iirt24 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt24 -- python $R/tools/exp/iir_stage_times.py 8 24 20 512 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt24 ) > $O/iir_launches_call_bpo24.txt 2>&1; cat $O/iir_launches_call_bpo24.txt
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > $O/iir_launches_call.txt 2>&1; grep "scan\|launches" $O/iir_launches_call.txt
