#!/bin/bash
# round 5, session y: the zero-state table product's loop for whole slices of aligned rows (all of a block's loads issued before the first
# conversion / MFMA; the general loop converted stage 0's float32 samples inside the branch that loaded them: a wait behind every load):
# parity, rates (shipped = 2 blocks per trip; variant z4 = 4), launch list
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_sharding_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for v in base z4; do
  for cfg in "--bpo 3 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 24 --log2-samples 20 --channels 8 --chunk 512" "--bpo 24 --log2-samples 20 --channels 64 --chunk 512" "--bpo 3 --log2-samples 22 --channels 2 --chunk 1024"; do
    echo -n "$v $cfg: "; ( [ $v != base ] && export FRT_LIB_VARIANT=$v; timeout 120 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (d['ms'], d['octave_bands_per_s']))" )
  done
done
done
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) 2>&1 | tail -32
