#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the gradient's vector gather (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gv_traffic
rm -rf $out; mkdir -p $out
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  PYTHONPATH=$R timeout -s KILL 180 rocprofv3 --pmc $grp --kernel-include-regex "k_gather_vector|k_local_reg" --output-format csv -d $out/p$i -- python3 $R/tools/exp/kernel_times.py tet2 ${1:-55} 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; exit 1; }
done
grep neohookean $out/p1.log
cd $R && python3 tools/pmc_summary.py gpurun_out/gv_traffic
