#!/bin/bash
# PMC passes over the row-assembly kernel (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/asm_pmc
rm -rf $out; mkdir -p $out
timeout -s KILL 60 rocprofv3 -L > $out/counters_list.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  PYTHONPATH=$R timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-include-regex "k_assemble_rows" --output-format csv -d $out/p$i -- python3 $R/tools/exp/asm_time.py 55 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; exit 1; }
done
cd $R && python3 tools/pmc_summary.py gpurun_out/asm_pmc
