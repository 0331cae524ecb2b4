#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of the row-assembly kernels for the default library and the variants named
# (run on the GPU box from the repo root): tools/exp/asm_traffic.sh <grid> [variant...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
grid=$1; shift
for v in default "$@"; do
  out=$R/gpurun_out/asm_traffic_$v
  rm -rf $out; mkdir -p $out
  [ $v = default ] && unset PFX_LIBRARY || export PFX_LIBRARY=$R/polyfem_amd/variants/libpfx_$v.so
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    PYTHONPATH=$R timeout -s KILL 180 rocprofv3 --pmc $grp --kernel-include-regex "k_assemble_rows" --output-format csv -d $out/p$i -- python3 $R/tools/exp/asm_time.py $grid 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; exit 1; }
  done
  echo "== $v"; grep "rows" $out/p1.log
  (cd $R && python3 tools/pmc_summary.py gpurun_out/asm_traffic_$v)
done
