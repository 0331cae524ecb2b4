#!/bin/bash
# PMC passes over the row-assembly kernel (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/asm_pmc
mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  PYTHONPATH=$R timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-include-regex "k_assemble_rows" --output-format csv -d $out/p$i -- python3 $R/tools/exp/asm_time.py 55 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; exit 1; }
done
cd $R && python3 tools/pmc_summary.py gpurun_out/asm_pmc
