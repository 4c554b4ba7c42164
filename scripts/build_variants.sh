#!/bin/bash
# Perf experiments: build libastrild_hip_<tag>.so with extra -D flags for ONE source file (default mesh_paint_tiled.hip),
# e.g.  scripts/build_variants.sh abl2 -DPAINT_ABLATE=2          scripts/build_variants.sh nt kappa.hip -DKSTACK_NT=1
# run with ASTRILD_HIP_LIB=astrild_amd/libastrild_hip_<tag>.so
set -e
cd "$(dirname "$0")/../astrild_amd/csrc"
tag=$1; shift
src=mesh_paint_tiled.hip
case "$1" in *.hip) src=$1; shift;; esac
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function "$@" -c $src -o ../variant_$tag.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../variant_$tag.o -o ../libastrild_hip_$tag.so -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
rm -f ../variant_$tag.o
echo built ../libastrild_hip_$tag.so
