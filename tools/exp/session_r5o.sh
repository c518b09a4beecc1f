#!/bin/bash
# round 5, session o: the exact bank's output pass in its lane-split form (band filter per lane on three lanes of a chunk, decimator on a
# pair): parity, timing against the filter-group-per-lane form (variant ix, FRT_LANE_SPLIT=0), launch list
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_iir_gpu.py tests/test_sharding_gpu.py tests/test_widgets_gpu.py -x -q 2>&1 | tail -4
for rep in 1 2; do
for mode in split plain; do
  if [ $mode = split ]; then unset FRT_LANE_SPLIT; else export FRT_LANE_SPLIT=0; fi
  for cfg in "--bpo 3 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 3 --log2-samples 22 --channels 8 --chunk 512" "--bpo 24 --log2-samples 20 --channels 8 --chunk 512" "--bpo 1 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 3 --log2-samples 22 --channels 2 --chunk 1024"; do
    echo -n "$mode $cfg: "; FRT_LIB_VARIANT=ix timeout 120 python tools/bench_octbank.py $cfg --iters 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (d['ms'], d['octave_bands_per_s']))"
  done
done
done
unset FRT_LANE_SPLIT
OUT=$R/gpurun_out/iir_split; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && FRT_LIB_VARIANT=ix timeout 200 rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python $R/tools/exp/iir_stage_times.py 8 3 22 > $OUT.log 2>&1 )
python tools/exp/iir_stage_times.py --parse $OUT
