#!/bin/bash
# round 5, session c: (1) the whole GPU suite on the library without environment switches (frt_set_option, split rows, ring fix);
# (2) packed vs split rows on ALIGNED buffer sets, three / four waves per SIMD (tables in LDS), ring instance for the colour kind;
# (3) write-request counters packed vs split; (4) parity of the four-wave build
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
echo "== (1) GPU suite"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
B=tools/bin/stft_selftest
S="s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//"
echo "== (2) A/B"
for rep in 1 2; do
for v in base bx t3 w4; do
  if [ $v = base ]; then LP=""; else LP=$R/tools/variants/$v; fi
  for cfg in "3 0 40 32 0" "3 0 40 32 1" "0 0 40 32 0" "0 0 40 32 1"; do
    echo -n "$v: "; LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
  done
  if [ $v != base ]; then
    for cfg in "3 0 40 32 0" "3 0 40 32 1"; do
      echo -n "$v ring-image: "; FRT_STFT_RING_IMAGE=1 LD_LIBRARY_PATH=$LP:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
    done
  fi
done
done
echo "== w4, run lengths"
for run in 8 12 16 24; do
  for cfg in "3 $run 40 32 1" "0 $run 40 32 1"; do
    echo -n "w4: "; LD_LIBRARY_PATH=$R/tools/variants/w4:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
    echo -n "w4 ring-image: "; FRT_STFT_RING_IMAGE=1 LD_LIBRARY_PATH=$R/tools/variants/w4:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "$S"
  done
done
echo "== (4) parity of the four-wave build (library swapped in the scratch copy)"
cp friture_amd/lib/libfriture_hip.so /tmp/base.so
cp tools/variants/w4/libfriture_hip.so friture_amd/lib/libfriture_hip.so
timeout 600 python -m pytest tests/test_stft_gpu.py -x -q 2>&1 | tail -3
FRT_STFT_RING_IMAGE=1 timeout 600 python -m pytest tests/test_stft_gpu.py -x -q -k "image or split or ring" 2>&1 | tail -3
cp /tmp/base.so friture_amd/lib/libfriture_hip.so
echo "== (3) write-request counters, packed vs split (colour kind), aligned sets"
OUT=$R/gpurun_out/pmc_r5c; rm -rf $OUT; mkdir -p $OUT
pass() { v=$1; name=$2; shift 2; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$v/$name -o p --output-format csv -- $CMD > $OUT/$v.$name.log 2>&1 ); echo "pass $v/$name rc=$?"; }
for v in packed split; do
  if [ $v = packed ]; then CMD="$R/tools/bin/stft_selftest bench 1024 512 1 26 3 0 5 32 0"; else CMD="$R/tools/bin/stft_selftest bench 1024 512 1 26 3 0 5 32 1"; fi
  pass $v tcp1 TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TCR_TCP_STALL_CYCLES
  pass $v tcp2 TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_GATE_EN1 TCP_GATE_EN2
  pass $v tcc1 TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL
  pass $v ta1 TA_TA_BUSY TA_TOTAL_WAVEFRONTS
  pass $v ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
  python $R/tools/prof_summary.py pmc $OUT/$v stft_kernel > $R/gpurun_out/r5c_pmc_$v.txt
done
