#!/bin/bash
# Library variant with extra macros for fft_fused.hip: tools/build_fft_variant.sh <tag> <-Dmacro ...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
python -m toast_amd.build > /dev/null
B=toast_amd/build
hipcc -x hip -c toast_amd/csrc/fft_fused.hip -o $B/fft_fused_$tag.o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@"
OBJS=$(ls $B/*.o | grep -v "fft_fused\|pybind")
hipcc -shared -fPIC --offload-arch=gfx950 -o $B/libtoast_hip_$tag.so $OBJS $B/fft_fused_$tag.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
echo $B/libtoast_hip_$tag.so
