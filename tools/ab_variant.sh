#!/bin/bash
# A/B helper: build a variant of the library that differs only in the compile-time knobs of trace_kernels.hip
# (e.g. -DTWK_TRACE_WAVES=7 -DTWK_TRACE_STACK_LDS=20) into build/lib_<name>.so; the other objects are reused.
# usage: tools/ab_variant.sh <name> <extra hipcc flags...>     then on the GPU box:
#        cp build/lib_<name>.so tweeker_raytracer_amd/libtweeker_hip.so && python bench.py --no-cpu-baseline
set -e
NAME=$1; shift
cd "$(dirname "$0")/../tweeker_raytracer_amd/csrc"
make -s > /dev/null
mkdir -p ../../build
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function "$@" -c trace_kernels.hip -o ../../build/trace_$NAME.o 2>&1 | grep -E "error" || true
hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/lib_$NAME.so device_api.o bvh_build.o ../../build/trace_$NAME.o shade_kernels.o tail_kernel.o host/description_parser.o host/triangle_meshes.o host/application.o host/image_files.o host/host_cabi.o -lz
ls -la ../../build/lib_$NAME.so
