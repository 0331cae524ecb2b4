#!/bin/bash
# usage (on the GPU box): tools/exp/rows_time.sh <tag> [bench flags]   -- average duration of the row-assembly kernel of one short bench run (kernel trace)
tag=$1; shift
out=gpurun_out/rt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-secondary --no-amg --steps 3 --warmup 1 --cg-max-iter 4 "$@" > $out/trace.log 2>&1 || { echo "trace failed"; tail -n 3 $out/trace.log; exit 1; }
echo "== $tag: $(find $out -name '*kernel_stats.csv' -exec grep -h k_assemble_rows {} \; | sed 's/"void [^"]*"/rows/' | cut -c1-48)"
find $out -name "*.csv" -size +1M -delete
