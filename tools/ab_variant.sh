#!/bin/bash
# A/B helper: build a variant of the library that differs only in the compile-time knobs of trace_kernels.hip
# (e.g. -DTWK_TRACE_WAVES=7 -DTWK_TRACE_STACK_LDS=20) into build/lib_<name>.so; the other objects are reused.
# usage: tools/ab_variant.sh <name> <extra hipcc flags...>     then on the GPU box:
#        TWK_LIB=build/lib_<name>.so python bench.py --no-cpu-baseline
# (flags that change device_types.h constants used by other objects, e.g. TWK_TRACE_STACK_LDS, rebuild those too)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../tweeker_raytracer_amd/csrc"
make -s > /dev/null
mkdir -p ../../build/$NAME
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-function"
for f in trace_kernels bvh_build bvh_sah device_api shade_kernels; do
  ( hipcc $FLAGS "$@" -c $f.hip -o ../../build/$NAME/$f.o 2>&1 | grep -E "error" || true ) &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/lib_$NAME.so ../../build/$NAME/device_api.o ../../build/$NAME/bvh_build.o ../../build/$NAME/bvh_sah.o ../../build/$NAME/trace_kernels.o ../../build/$NAME/shade_kernels.o host/description_parser.o host/triangle_meshes.o host/application.o host/image_files.o host/host_cabi.o -lz
ls -la ../../build/lib_$NAME.so
