#!/bin/bash
# A/B builds of the library: tools/ab/build_variant.sh <tag> "<-D flags>"
# -> opencl-raytracer_amd/csrc/variants/libhip_raytracer_<tag>.so (git-ignored, travels with gpurun).
# Load one with RT_LIB_OVERRIDE=<path> (hip_raytracer.load_library).
set -e
cd "$(dirname "$0")/../../opencl-raytracer_amd/csrc"
tag=$1; defs=$2
mkdir -p variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -I../../include -I. $defs"
for f in rt_kernels rt_wavefront; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o variants/${f}_$tag.o & done
/opt/rocm/bin/hipcc $FLAGS -x hip -c rt_api.cpp -o variants/rt_api_$tag.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 variants/rt_kernels_$tag.o variants/rt_wavefront_$tag.o variants/rt_api_$tag.o -o variants/libhip_raytracer_$tag.so
rm -f variants/*_$tag.o
echo built $tag
