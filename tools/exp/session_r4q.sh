#!/bin/bash
# Round 4, energy recurrence of the banks: pipelined trips + no carry launch (shipped) against the previous kernels (head) and a chain trip of 8 (c8).
#   gpurun --timeout 400 -- 'bash tools/exp/session_r4q.sh'
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity: banks"; timeout 300 python -m pytest tests/test_iir_gpu.py tests/test_ola_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in "" head; do
  for bpo in 3 24; do
    echo -n "variant '${v:-shipped}' bpo $bpo: "; FRT_LIB_VARIANT=$v timeout 120 python tools/bench_octbank.py --chunk 1024 --bpo $bpo --iters 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (r['ms'], r['octave_bands_per_s']))"
  done
done
echo "== launches of one call (shipped)"
( cd /tmp && rm -rf /tmp/iirt && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/iirt -- python $R/tools/exp/iir_stage_times.py 8 3 22 > /dev/null 2>&1; python $R/tools/exp/iir_stage_times.py --parse /tmp/iirt ) > gpurun_out/r04_iir_launches.txt 2>&1; tail -8 gpurun_out/r04_iir_launches.txt
