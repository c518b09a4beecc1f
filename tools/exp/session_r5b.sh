#!/bin/bash
# round 5, session b: (1) where split rows differ from packed rows at full size; (2) run-length sweep of the headline kernel
# (powers of two against odd lengths: do the waves walk HBM channels in lockstep?); (3) the short-block energy tests of the
# exact bank; (4) write-request counters packed vs split; (5) lane_probe; (6) a full bench line
set -u
R=$GRAFT_REPO_ROOT; cd $R
export TMPDIR=/tmp FRT_BENCH_SETS=4
echo "== (1) split debug"; timeout 300 python tools/exp/split_debug.py 2>&1 | tail -30
echo "== (3) exact bank, short energy blocks"; timeout 600 python -m pytest tests/test_iir_gpu.py -x -q 2>&1 | tail -5
B=tools/bin/stft_selftest
echo "== (2) run-length sweep, shipped library"
for run in 8 11 12 13 15 16 17 19 21 23 24 27 32; do
  for cfg in "3 $run 40 32 0" "3 $run 40 32 1" "0 $run 40 32 1"; do
    echo -n "base: "; timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//"
  done
done
echo "== nt variant"
for run in 13 16 19; do
  for cfg in "3 $run 40 32 0" "3 $run 40 32 1"; do
    echo -n "nt: "; LD_LIBRARY_PATH=$R/tools/variants/nt:${LD_LIBRARY_PATH:-} timeout 120 $B bench 1024 512 1 26 $cfg | tail -1 | sed "s/bench p32 N=1024 hop=512 C=1 T=2^26 F=131071 //; s/algorithmic.*of 8 TB.s)//"
  done
done
echo "== (5) lane probe"; timeout 120 tools/exp/lane_probe.bin
echo "== (4) write-request counters, packed vs split (colour kind)"
OUT=$R/gpurun_out/pmc_r5b; rm -rf $OUT; mkdir -p $OUT
pass() { v=$1; name=$2; shift 2; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$v/$name -o p --output-format csv -- $CMD > $OUT/$v.$name.log 2>&1 ); echo "pass $v/$name rc=$?"; }
for v in packed split; do
  if [ $v = packed ]; then CMD="$R/tools/bin/stft_selftest bench 1024 512 1 26 3 0 5 32 0"; else CMD="$R/tools/bin/stft_selftest bench 1024 512 1 26 3 0 5 32 1"; fi
  pass $v tcp1 TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TCR_TCP_STALL_CYCLES
  pass $v tcp2 TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_GATE_EN1 TCP_GATE_EN2
  pass $v tcc1 TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL
  pass $v ta1 TA_TA_BUSY TA_TOTAL_WAVEFRONTS
  pass $v ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
  pass $v sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  python $R/tools/prof_summary.py pmc $OUT/$v stft_kernel > $R/gpurun_out/r5b_pmc_$v.txt
done
echo "== (6) bench line"
timeout 600 python bench.py --full-json gpurun_out/r5b_bench_full.json > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err; echo "bench rc=$?"; wc -c gpurun_out/r5b_bench.json; tail -3 gpurun_out/r5b_bench.err
