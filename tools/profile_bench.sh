#!/bin/bash
# usage (on the GPU box): tools/profile_bench.sh <tag> [bench flags...]
# 1) kernel trace + stats of the bench command (default workload, the secondary configuration left out so that the
#    per-kernel averages belong to one problem size), 2) FETCH_SIZE and WRITE_SIZE passes (one run each) on a short run of
#    the same workload.  Results under gpurun_out/<tag>/; kernel_stats.csv, pmc_summary.txt and pmc_traffic.json are the
#    files to copy into profiles/.
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > $out/trace.log 2>&1 || { echo "trace run failed"; tail -n 5 $out/trace.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-include-regex "k_sell_spmv|k_cg_|k_assemble_rows|k_local_|k_element_geometry|k_sell_values|k_gather_vector" --output-format csv -d $out/pmc_$c -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --cg-max-iter 24 "$@" > $out/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -n 5 $out/pmc_$c.log; exit 1; }
done
python3 tools/pmc_summary.py $out $out/pmc_traffic.json > $out/pmc_summary.txt
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*.csv" -size +2M -delete
grep -h "k_sell_spmv\|k_assemble_rows\|k_sell_values\|k_local_reg" $out/kernel_stats.csv | cut -c1-200
