#!/bin/bash
# round 6, session d: the chunk scan on the matrix cores (parity; timelines with and without the stage-0 look-back), OLA advisor fixes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== iir + ola tests"; timeout 1500 python -m pytest tests/test_iir_gpu.py tests/test_ola_gpu.py tests/test_soak_gpu.py -x -q 2>&1 | tail -5
echo "== bank times (shipped: mfma scan + look-back at stage 0)"
for cfg in "--bpo 3 --log2-samples 22 --chunk 1024" "--bpo 3 --log2-samples 22 --chunk 512" "--bpo 24 --log2-samples 20 --chunk 512" "--bpo 24 --log2-samples 20 --chunk 1024"; do
  timeout 300 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['bpo'], r['chunk'], round(r['ms'],4), 'ms')"
done 2>&1 | tee $O/bank_times.txt
echo "== variant look0 (mfma scan at every stage, no look-back)"
for cfg in "--bpo 3 --log2-samples 22 --chunk 1024" "--bpo 3 --log2-samples 22 --chunk 512" "--bpo 24 --log2-samples 20 --chunk 512"; do
  FRT_LIB_VARIANT=look0 timeout 300 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['bpo'], r['chunk'], round(r['ms'],4), 'ms')"
done 2>&1 | tee $O/bank_times_look0.txt
echo "== launches of one call, shipped library, 8 ch x 27 bands, chunks of 1024"
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > $O/iir_launches_call.txt 2>&1; cat $O/iir_launches_call.txt
echo "== the same, variant look0"
( cd /tmp && rm -rf /tmp/iirt2 && FRT_LIB_VARIANT=look0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt2 -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt2 ) > $O/iir_launches_call_look0.txt 2>&1; tail -30 $O/iir_launches_call_look0.txt
echo "== 216 bands, chunks of 512"
( cd /tmp && rm -rf /tmp/iirt24 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt24 -- python $R/tools/exp/iir_stage_times.py 8 24 20 512 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt24 ) > $O/iir_launches_call_bpo24.txt 2>&1; tail -32 $O/iir_launches_call_bpo24.txt
