#!/bin/bash
# PMC passes for one STFT configuration.  usage: gpu_pmc.sh <tag> <run (negative = generic kernel)> [kind] [N hop C log2T] [passes]
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-wave}; RUN=${2:-16}; KIND=${3:-0}; N=${4:-1024}; HOP=${5:-512}; CH=${6:-1}; LT=${7:-26}; PASSES=${8:-all}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="$R/tools/bin/stft_selftest bench $N $HOP $CH $LT $KIND $RUN 5"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- $CMD > $OUT/$name.log 2>&1; echo "pass $TAG/$name rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
[ "$PASSES" = "sq" ] && exit 0
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_WRITE_sum
