#!/bin/bash
# Short GPU-box session after a kernel change outside the headline path: parity tests, per-kernel table, pitch and
# big-kernel rocprofv3 stats.  Outputs -> gpurun_out/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench_all"; timeout 900 python tools/bench_all.py > gpurun_out/bench_all.json 2> gpurun_out/bench_all.err; wc -l gpurun_out/bench_all.json
echo "== rocprofv3 pitch"; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/pitch -o pitch -- python $R/tools/prof_pitch.py > $R/gpurun_out/prof/pitch.log 2>&1 ); echo rc=$?
echo "== rocprofv3 N=2048/4096/16384 image"; for n in "2048 1024 8 24" "4096 1024 16 22" "16384 8192 32 20"; do
  tag=$(echo $n | cut -d' ' -f1)
  ( cd /tmp && FRT_BENCH_SETS=4 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/big$tag -o big -- $R/tools/bin/stft_selftest bench $n 3 0 30 > $R/gpurun_out/prof/big$tag.log 2>&1 ); echo "$tag rc=$?"; tail -1 $R/gpurun_out/prof/big$tag.log
done
