#!/bin/bash
# round 5, session x: the zero-state table product of the time-parallel IIR bank with CG = 1 / 2 / 4 groups of 16 columns per wavefront
# (every table value then feeds CG MFMAs; with one group every wavefront reads the stage's whole table: 4x the bytes of its samples):
# parity (shipped rule), rates with CG forced on the -DFRT_EXPERIMENTS build (variant ix, FRT_ZS_CG), launch list of one call
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_sharding_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for cg in 1 2 4; do
  for cfg in "--bpo 3 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 24 --log2-samples 20 --channels 8 --chunk 512" "--bpo 24 --log2-samples 20 --channels 64 --chunk 512" "--bpo 3 --log2-samples 22 --channels 2 --chunk 1024"; do
    echo -n "CG=$cg $cfg: "; FRT_ZS_CG=$cg FRT_LIB_VARIANT=ix timeout 120 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (d['ms'], d['octave_bands_per_s']))"
  done
done
done
echo "== shipped rule"
for cfg in "--bpo 3 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 24 --log2-samples 20 --channels 8 --chunk 512"; do
  echo -n "shipped $cfg: "; timeout 120 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (d['ms'], d['octave_bands_per_s']))"
done
( cd /tmp && rm -rf /tmp/iirt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) 2>&1 | tail -32
