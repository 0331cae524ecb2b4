#!/bin/bash
# usage (on the GPU box): tools/exp/placement_pmc.sh
# The CG product at 10 M tets in the two placements that PFX_EXPERIMENT_NO_ORDER selects (identical virtual addresses, different
# physical pages), one rocprofv3 --pmc pass per counter group and placement: translation (UTCL1) and L2 / memory-path counters.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/placement
pass() { # tag counters...
  tag=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex "k_sell_spmv" --output-format csv -d gpurun_out/placement/$tag -- python3 bench.py --no-cpu-baseline --no-secondary --no-amg --steps 1 --warmup 0 --cg-max-iter 40 > gpurun_out/placement/$tag.log 2>&1 || { echo "$tag failed"; tail -n 3 gpurun_out/placement/$tag.log; exit 1; }
  python3 tools/exp/bimodal_summary.py gpurun_out/placement/$tag $tag | grep "true,true,double"
  find gpurun_out/placement/$tag -name "*.csv" -size +2M -delete
}
for mode in order noorder; do
  if [ $mode = noorder ]; then export PFX_EXPERIMENT_NO_ORDER=1; else unset PFX_EXPERIMENT_NO_ORDER; fi
  pass ${mode}_tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE || exit 1
  pass ${mode}_tcc TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 1
  pass ${mode}_tcp TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE || exit 1
done
