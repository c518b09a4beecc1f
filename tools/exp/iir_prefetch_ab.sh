#!/bin/bash
# the lane passes' sample prefetch, A/B: in-tree library, tools/variants/<names>: launch sequence of one call (27 and 216 bands)
set -u
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/iir_prefetch; mkdir -p $O
timeout 900 python -m pytest tests/test_iir_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for v in intree "$@"; do
  for cfg in "8 3 22 1024" "8 24 20 512"; do
    tag=$(echo $cfg | tr ' ' '_'); D=/tmp/iirt_${v}_$tag; rm -rf $D
    if [ "$v" = intree ]; then ( cd /tmp && env -u FRT_LIB_VARIANT timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/tools/exp/iir_stage_times.py $cfg > /dev/null 2>&1 )
    else ( cd /tmp && FRT_LIB_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/tools/exp/iir_stage_times.py $cfg > /dev/null 2>&1 ); fi
    python tools/exp/iir_stage_times.py --parse $D > $O/${v}_$tag.txt 2>&1
    echo "== $v $cfg"; grep "lane\|launches" $O/${v}_$tag.txt
  done
done
