#!/bin/bash
# Perf experiments: build libastrild_hip_<tag>.so with extra -D flags for mesh_paint_tiled.hip,
# e.g.  scripts/build_variants.sh abl2 -DPAINT_ABLATE=2 ; run with ASTRILD_HIP_LIB=astrild_amd/libastrild_hip_abl2.so
set -e
cd "$(dirname "$0")/../astrild_amd/csrc"
tag=$1; shift
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function "$@" -c mesh_paint_tiled.hip -o ../variant_$tag.o
objs=$(ls *.o | grep -v mesh_paint_tiled.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../variant_$tag.o -o ../libastrild_hip_$tag.so -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
rm -f ../variant_$tag.o
echo built ../libastrild_hip_$tag.so
