#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ola_gpu.py tests/test_widgets_gpu.py -x -q -m gpu 2>&1 | tail -3
for cfg in "8 3 22 10" "8 24 20 5"; do python tools/exp/fir_only.py $cfg; done
cd /tmp
for v in base; do
  for cfg in "8 3 22 3" "8 24 20 3"; do
    tag=${v}_$(echo $cfg | tr ' ' '_')
    rocprofv3 --kernel-trace -d $R/gpurun_out/fir_trace/$tag -o p --output-format csv -- python $R/tools/exp/fir_only.py $cfg > /dev/null 2>&1
  done
done
cd $R; python tools/exp/stage_times.py gpurun_out/fir_trace
