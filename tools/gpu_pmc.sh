#!/bin/bash
# PMC passes + ablation timings for the N=1024 STFT kernel.  Outputs -> gpurun_out/pmc/.
set -u
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
KIND=${KIND:-3}
CMD="$R/tools/bin/stft_selftest bench 1024 512 1 26 $KIND 16 5"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc/$name -o p --output-format csv -- $CMD > $R/gpurun_out/pmc/$name.log 2>&1; echo "pass $name rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
pass sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_ADD_F32
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_WRITE_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "== ablation (LD_LIBRARY_PATH -> ablate build)"
export LD_LIBRARY_PATH=$R/friture_amd/lib/ablate
for kind in 0 3; do for ab in 0 1 2 3 4 8 12 5 6 7 15; do echo -n "kind=$kind ablate=$ab: "; FRT_ABLATE=$ab timeout 60 $R/tools/bin/stft_selftest bench 1024 512 1 26 $kind 16 20 | tail -1; done; done
