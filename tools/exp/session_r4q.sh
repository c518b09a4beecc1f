#!/bin/bash
# SQ counters of the batched overlap-add bank's kernel (two passes), 8 ch x 2^22, bpo 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r4q
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/tools/exp/ola_stage_times.py 8 3 22"
pass() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- $CMD > $OUT/$name.log 2>&1 ); echo "pass $name rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python $R/tools/prof_summary.py pmc $OUT ola_pair | tee $R/gpurun_out/r04_ola_pair_pmc.txt
