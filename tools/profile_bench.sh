#!/bin/bash
# usage (on the GPU box): tools/profile_bench.sh <tag>
# 1) kernel trace + stats of the default bench command, 2) FETCH_SIZE and WRITE_SIZE passes (one run each) on a short run
# of the same workload.  Results under gpurun_out/<tag>/.
set -e
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex "k_sell_spmv|k_cg_|k_assemble_rows|k_local_|k_element_geometry|k_sell_values|k_finish" --output-format csv -d $out/pmc_$c -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --cg-max-iter 24 > $out/pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*.csv" -size +2M -delete
