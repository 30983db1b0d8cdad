#!/bin/bash
# Experimental library with phase clocks in the FFT row pass: toast_amd/build/libtoast_hip_phase.so
# (use with TOAST_HIP_LIBRARY=... tools/exp_fft_phases.py).  The other objects come from the normal build.
set -e
cd "$(dirname "$0")/.."
python -m toast_amd.build > /dev/null
B=toast_amd/build
hipcc -x hip -c toast_amd/csrc/fft_fused.hip -o $B/fft_fused_phase.o --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
    -ffp-contract=off -DTOAST_FFT_PHASE_CLOCK
OBJS=$(ls $B/*.o | grep -v "fft_fused\|pybind")
hipcc -shared -fPIC --offload-arch=gfx950 -o $B/libtoast_hip_phase.so $OBJS $B/fft_fused_phase.o -L/opt/rocm/lib -lrocfft \
    -Wl,-rpath,/opt/rocm/lib
echo $B/libtoast_hip_phase.so
