#!/bin/bash
# usage (GPU box): tools/exp/run_variants.sh <name>...  -> assembly timings of the default library and of each variant
echo "== default"; timeout -k 10 200 python3 tools/exp/asm_time.py 55 10 || exit 1
for v in "$@"; do
  echo "== $v"; PFX_LIBRARY=$PWD/polyfem_amd/variants/libpfx_$v.so timeout -k 10 200 python3 tools/exp/asm_time.py 55 10 || exit 1
done
