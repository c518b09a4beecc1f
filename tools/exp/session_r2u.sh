#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; export FRT_BENCH_SETS=4; cd /tmp
for v in window ring; do
  for kind in 0 3; do
    if [ $v = ring ]; then export FRT_STFT_RING_IMAGE=1; unset FRT_STFT_NO_RING; else export FRT_STFT_NO_RING=1; unset FRT_STFT_RING_IMAGE; fi
    OUT=$R/gpurun_out/pmc_ringcmp/${v}_k$kind; mkdir -p $OUT
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $OUT/sq1 -o p --output-format csv -- $R/tools/bin/stft_selftest bench 1024 512 1 26 $kind 0 5 > /dev/null 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY -d $OUT/sq2 -o p --output-format csv -- $R/tools/bin/stft_selftest bench 1024 512 1 26 $kind 0 5 > /dev/null 2>&1
  done
done
cd $R
for v in window ring; do for kind in 0 3; do echo "== $v kind $kind"; python tools/prof_summary.py pmc gpurun_out/pmc_ringcmp/${v}_k$kind stft_kernel | grep -v "^#" | awk '{print $2, $4}' | tr '\n' ';'; echo; done; done
