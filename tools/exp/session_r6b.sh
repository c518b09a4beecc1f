#!/bin/bash
# round 6, session b: the exact IIR bank's look-back output pass (parity; per-launch timelines and bank times at look-back limits 0 / 6 / 12 / 24;
# chunk sweep with the look-back on)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== iir tests"; timeout 1500 python -m pytest tests/test_iir_gpu.py -x -q 2>&1 | tail -5
for v in look0 look6 "" look24; do
  echo "== bank times, variant '${v:-shipped (12)}': 8 ch x 27 bands chunk 1024 / 512; 8 ch x 216 bands chunk 512"
  for cfg in "--bpo 3 --log2-samples 22 --chunk 1024" "--bpo 3 --log2-samples 22 --chunk 512" "--bpo 24 --log2-samples 20 --chunk 512" "--bpo 24 --log2-samples 20 --chunk 1024"; do
    FRT_LIB_VARIANT=$v timeout 300 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['bpo'], r['chunk'], round(r['ms'],4), 'ms')"
  done
done 2>&1 | tee $O/bank_times.txt
echo "== launches of one call, shipped library, 8 ch x 27 bands, chunks of 1024"
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > $O/iir_launches_call.txt 2>&1; cat $O/iir_launches_call.txt
echo "== the same with the look-back limit 24"
( cd /tmp && rm -rf /tmp/iirt2 && FRT_LIB_VARIANT=look24 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt2 -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt2 ) > $O/iir_launches_call_look24.txt 2>&1; cat $O/iir_launches_call_look24.txt
