#!/bin/bash
# usage: scripts/build_variant.sh NAME [extra hipcc flags...]
# Builds scripts/_build/librt_NAME.so: the library with kernels.hip compiled with the extra flags
# (experiments; the product is built by ray_tracing_amd/csrc/Makefile).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../ray_tracing_amd/csrc"
make -s
OUT=../../scripts/_build
mkdir -p $OUT
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 "$@" -c kernels.hip -o $OUT/kernels_$NAME.o
OBJS=$(ls _build/*.o | grep -v kernels.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/librt_$NAME.so $OUT/kernels_$NAME.o $OBJS -lpthread
echo built $OUT/librt_$NAME.so
