#!/bin/bash
# Builds path_tracer_amd/_ab/libptcore_<name>.so: the library with extra compiler
# flags (-DPTC_...=...) on the render kernels' sources, for tools/ab.sh.
# Usage: tools/build_variant.sh <name> [flags...]
set -eu
cd "$(dirname "$0")/../path_tracer_amd/csrc"
name=$1; shift
mkdir -p ../_ab/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result"
for f in ptc_wavefront ptc_render; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o ../_ab/obj_$name/$f.o &
done
wait
make -s ptc_api.o ptc_bvh_build.o ptc_image_io.o ptc_mesh_io.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_ab/libptcore_$name.so ../_ab/obj_$name/ptc_wavefront.o ../_ab/obj_$name/ptc_render.o ptc_api.o ptc_bvh_build.o ptc_image_io.o ptc_mesh_io.o
echo built ../_ab/libptcore_$name.so
