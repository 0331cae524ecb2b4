#!/bin/bash
# usage (GPU box): tools/quick_bench.sh  -> GPU tests, then short linear and NeoHookean bench runs (assembly timings)
set -o pipefail
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for w in linear_elasticity neohookean; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --cg-max-iter 50 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1 || { tail -5 gpurun_out/bench_$w.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$w.log").read().strip().splitlines()[-1])
a=d["assembly"]; print("$w", "local_ms %.3f rows_ms %.3f convert_ms %.3f" % (a["local_ms"], a["gather_ms"], a["solver_copy_ms"]), "spmv_ms %.4f" % d["roofline"]["avg_ms"], "ms_per_step %.2f" % d["ms_per_step"])
PY
done
