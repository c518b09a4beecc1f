#!/bin/bash
# round 4, session i: which unit holds the headline kernel (N = 1024, hop 512, colour kind) — memory-path counters of the
# shipped kernel against the store-less ablation (-DFRT_ABLATE build, FRT_ABLATE=1) and the load-less one (FRT_ABLATE=2)
set -u
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_r4i
rm -rf $OUT; mkdir -p $OUT
CMD="$R/tools/bin/stft_selftest bench 1024 512 1 26 3 0 5"
pass() { v=$1; name=$2; shift 2; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$v/$name -o p --output-format csv -- $CMD > $OUT/$v.$name.log 2>&1 ); echo "pass $v/$name rc=$?"; }
for v in shipped nostores noloads; do
  case $v in
    shipped) export LD_LIBRARY_PATH=$R/tools/variants/abl; export FRT_ABLATE=0;;
    nostores) export LD_LIBRARY_PATH=$R/tools/variants/abl; export FRT_ABLATE=1;;
    noloads) export LD_LIBRARY_PATH=$R/tools/variants/abl; export FRT_ABLATE=2;;
  esac
  echo "== $v: $($CMD | tail -1 | cut -c1-140)"
  pass $v sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  pass $v sq2 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_FLAT
  pass $v ta1 TA_TA_BUSY TA_TOTAL_WAVEFRONTS
  pass $v ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
  pass $v ta3 TA_ADDR_STALLED_BY_TD_CYCLES TA_FLAT_WRITE_WAVEFRONTS
  pass $v td1 TD_TD_BUSY TD_TC_STALL
  pass $v td2 TD_STORE_WAVEFRONT TD_SPI_STALL
  pass $v tcp1 TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TCR_TCP_STALL_CYCLES
  pass $v tcp2 TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_GATE_EN1 TCP_GATE_EN2
  pass $v tcp3 TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCP_TA_ADDR_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES
  pass $v tcp4 TCP_RFIFO_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST
  pass $v tcc1 TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL
  pass $v tcc2 TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ_LEVEL TCC_REQ_sum
  pass $v grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY
  python $R/tools/prof_summary.py pmc $OUT/$v stft_kernel > $R/gpurun_out/r4i_$v.txt
done
unset LD_LIBRARY_PATH FRT_ABLATE
python - <<'PY'
import re
rows={}
for v in ("shipped","nostores","noloads"):
    for ln in open(f"gpurun_out/r4i_{v}.txt"):
        m=re.match(r"(\S+)\s+(\S+)\s+dispatches=\s*(\d+)\s+mean=(\S+)",ln)
        if m: rows.setdefault(m.group(2),{})[v]=float(m.group(4))
print("%-36s %14s %14s %14s"%("counter (mean per dispatch)","shipped","no stores","no loads"))
for k in sorted(rows): print("%-36s %14.6g %14.6g %14.6g"%(k,rows[k].get("shipped",float('nan')),rows[k].get("nostores",float('nan')),rows[k].get("noloads",float('nan'))))
PY
