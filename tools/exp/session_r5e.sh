#!/bin/bash
# round 5, session e: (1) the exact bank's output pass as workgroups of four wavefronts pinned one workgroup per CU (variant ix,
# FRT_LANE_WG4) against the single-wavefront workgroups; (2) split rows with non-temporal stores as shipped; (3) bench line
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
for mode in plain wg4; do
  OUT=$R/gpurun_out/iir_wg4_$mode; rm -rf $OUT; mkdir -p $OUT
  if [ $mode = plain ]; then unset FRT_LANE_WG4; else export FRT_LANE_WG4=1; fi
  ( cd /tmp && FRT_LIB_VARIANT=ix timeout 200 rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python $R/tools/exp/iir_stage_times.py 8 3 22 > $OUT.log 2>&1 )
  echo "== lane kernel: $mode"; python tools/exp/iir_stage_times.py --parse $OUT | grep -E "lane|launches"
  for cfg in "--bpo 3 --log2-samples 22 --channels 8 --chunk 1024" "--bpo 24 --log2-samples 20 --channels 8 --chunk 512"; do
    echo -n "$mode $cfg: "; FRT_LIB_VARIANT=ix timeout 120 python tools/bench_octbank.py $cfg --iters 20 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms  %.3e octave-bands/s' % (d['ms'], d['octave_bands_per_s']))"
  done
done
unset FRT_LANE_WG4
echo "== parity with the wg4 launches (variant swapped in)"
cp friture_amd/lib/libfriture_hip.so /tmp/base.so; cp tools/variants/ix/libfriture_hip.so friture_amd/lib/libfriture_hip.so
FRT_LANE_WG4=1 timeout 600 python -m pytest tests/test_iir_gpu.py -x -q 2>&1 | tail -3
cp /tmp/base.so friture_amd/lib/libfriture_hip.so
B=tools/bin/stft_selftest
S="s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//; s/\[isolated.*//"
for rep in 1 2; do
  for cfg in "3 0 40 32 0" "3 0 40 32 1" "0 0 40 32 0" "0 0 40 32 1"; do
    echo -n "base: "; timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
  done
done
timeout 300 python -m pytest tests/test_stft_gpu.py -x -q -k "split or ring" 2>&1 | tail -3
timeout 600 python bench.py --full-json gpurun_out/r5e_bench_full.json > gpurun_out/r5e_bench.json 2> gpurun_out/r5e_bench.err; echo "bench rc=$?"; wc -c gpurun_out/r5e_bench.json; tail -3 gpurun_out/r5e_bench.err
