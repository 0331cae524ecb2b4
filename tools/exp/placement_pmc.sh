#!/bin/bash
# usage (on the GPU box): tools/exp/placement_pmc.sh
# The CG product at 10 M tets with the placement search of the value array off (PFX_PLACEMENT_TRIES=0: whatever pages the
# first allocation got) and on (the fastest of up to six allocations), one rocprofv3 --pmc pass per counter group and setting:
# address translation (UTCL1), L2 and memory-path counters of k_sell_spmv<3,true,true,double>.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/placement
pass() { # tag counters...
  tag=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex "k_sell_spmv" --output-format csv -d gpurun_out/placement/$tag -- python3 bench.py --no-cpu-baseline --no-secondary --no-amg --steps 1 --warmup 0 --cg-max-iter 40 > gpurun_out/placement/$tag.log 2>&1 || { echo "$tag failed"; tail -n 3 gpurun_out/placement/$tag.log; exit 1; }
  python3 tools/exp/bimodal_summary.py gpurun_out/placement/$tag $tag | grep "true,true,double"
  find gpurun_out/placement/$tag -name "*.csv" -size +2M -delete
}
for tries in 0 6; do
  export PFX_PLACEMENT_TRIES=$tries
  pass search${tries}_tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE || exit 1
  pass search${tries}_tcc TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 1
  pass search${tries}_tcp TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE || exit 1
  pass search${tries}_ea TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 1
done
