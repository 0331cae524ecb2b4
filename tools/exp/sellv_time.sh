#!/bin/bash
# usage (on the GPU box): tools/exp/sellv_time.sh <variant> -- duration of k_sell_values<3,..,double> in one short bench run with PFX_SELLV=<variant>
v=$1
out=gpurun_out/sv_$v
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PFX_SELLV=$v; export PFX_SELL_ROWS3=${ROWS3:-0}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-secondary --no-amg --steps 3 --warmup 1 --cg-max-iter 4 > $out/trace.log 2>&1 || { echo "trace failed"; tail -n 3 $out/trace.log; exit 1; }
echo "== variant $v: $(find $out -name '*kernel_stats.csv' -exec grep -h 'k_sell_values<3' {} \; | sed 's/(long const[^"]*"//' | cut -c1-140)"
find $out -name "*.csv" -size +1M -delete
