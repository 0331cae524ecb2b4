#!/bin/bash
# usage: tools/exp/build_variant.sh <name> <extra hipcc flags...>   -> polyfem_amd/variants/libpfx_<name>.so (select with PFX_LIBRARY)
# Only the two files that instantiate the row assembly are rebuilt with the extra flags; the rest comes from `make`.
set -e
name=$1; shift
cd "$(dirname "$0")/../../polyfem_amd/csrc"
make -s -j8
b=/tmp/pfx_variant_$name
mkdir -p $b ../variants
COMMON="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-result $*"
/opt/rocm/bin/hipcc $COMMON -c pfx_local_nl.hip -o $b/pfx_local_nl.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -c pfx_local_linear.hip -o $b/pfx_local_linear.o &
wait
others=$(ls *.o | grep -v -e pfx_local_nl.o -e pfx_local_linear.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libpfx_$name.so $others $b/*.o -L/opt/rocm/lib -lrccl
echo built variants/libpfx_$name.so
