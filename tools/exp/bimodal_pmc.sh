#!/bin/bash
# usage (on the GPU box): tools/exp/bimodal_pmc.sh <n_processes>
# The same AMG solve at 10 M tets in <n> processes, one rocprofv3 --pmc pass each (L2 / memory-path counters of k_sell_spmv):
# per process the duration and the counters of the level-0 products, to compare a fast with a slow process.
n=${1:-4}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/bimodal
for k in $(seq 1 $n); do
  timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --kernel-include-regex "k_sell_spmv" --output-format csv -d gpurun_out/bimodal/p$k -- python3 tools/exp/amg_setup_time.py 119 > gpurun_out/bimodal/p$k.log 2>&1 || { echo "process $k failed"; tail -n 3 gpurun_out/bimodal/p$k.log; exit 1; }
  python3 tools/exp/bimodal_summary.py gpurun_out/bimodal/p$k $k
  find gpurun_out/bimodal/p$k -name "*.csv" -size +2M -delete
done
