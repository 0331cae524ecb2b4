#!/bin/bash
# usage (on the GPU box): tools/exp/rows_traffic.sh <tag> [bench flags]   -- time and FETCH_SIZE / WRITE_SIZE of the row-assembly kernel of one bench run
tag=$1; shift
out=gpurun_out/sv_traffic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-secondary --no-amg --steps 2 --warmup 1 --cg-max-iter 8 "$@" > $out/trace.log 2>&1 || { echo "trace failed"; tail -n 3 $out/trace.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-include-regex "k_sell_values" --output-format csv -d $out/pmc_$c -- python3 bench.py --no-cpu-baseline --no-secondary --no-amg --steps 1 --warmup 0 --cg-max-iter 8 "$@" > $out/pmc_$c.log 2>&1 || { echo "pmc $c failed"; exit 1; }
done
python3 tools/pmc_summary.py $out $out/pmc_traffic.json > $out/pmc_summary.txt
echo "== $tag: $(find $out -name '*kernel_stats.csv' -exec grep -h k_sell_values {} \; | sed 's/"void [^"]*"/sv/' | cut -c1-60)"
grep -A2 "k_sell_values" $out/pmc_summary.txt | grep SIZE
find $out -name "*.csv" -size +1M -delete
