#!/bin/bash
# Round 4, exact IIR bank: A/B of the shipped library against the previous commit's iir.hip (tools/variants/head), parity first.
#   gpurun --timeout 400 -- 'bash tools/exp/session_r4q.sh'
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity: all GPU tests"; timeout 300 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in "" d1 d4; do
  for cfg in "8 3 22" "8 24 20" "64 24 20"; do
    set -- $cfg
    echo -n "variant '${v:-shipped}' ch $1 bpo $2 2^$3: "; FRT_LIB_VARIANT=$v timeout 120 python tools/bench_octbank.py --chunk 1024 --channels $1 --bpo $2 --log2-samples $3 --iters 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (r['ms'], r['octave_bands_per_s']))"
  done
done
echo "== launches of one call (shipped)"
( cd /tmp && rm -rf /tmp/iirt && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > gpurun_out/r04_iir_launches.txt 2>&1; cat gpurun_out/r04_iir_launches.txt | cut -c1-110
