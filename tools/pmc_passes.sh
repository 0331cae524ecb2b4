#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <kernel-regex> -- <program...>   (run on the GPU box; one rocprofv3 --pmc pass per counter group)
out=$1; regex=$2; shift 3
mkdir -p gpurun_out/$out
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-include-regex "$regex" --output-format csv -d gpurun_out/$out/p$i -- "$@" > gpurun_out/$out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/$out/p$i.log; exit 1; }
done <<GROUPS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_BUSY_sum
GROUPS
python3 tools/pmc_summary.py gpurun_out/$out
