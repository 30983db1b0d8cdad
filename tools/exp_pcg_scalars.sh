#!/bin/bash
# PCG iteration time with the scalars on the host (three round trips per iteration) and on the device
# (csrc/pcg.hip), at configs[1] size (64 x 360 000 @ 100 Hz, Nside 512, 30 iterations) and at cfg-3.
# Run on the GPU box (profiles/r03_c).
out=${1:-gpurun_out/r03c/pcg_scalars.txt}
mkdir -p $(dirname $out); : > $out
for mode in host device host device; do
  echo "== TOAST_HIP_PCG_SCALARS=$mode  configs[1] size" >> $out
  TOAST_HIP_PCG_SCALARS=$mode python workflows/mapmaker_pcg.py --ndet 64 --minutes 60 --rate 100 --nside 512 --iter 30 --no-filter 2>&1 | grep -i "iteration\|MapMaker\|PCG" >> $out
done
for mode in host device; do
  echo "== TOAST_HIP_PCG_SCALARS=$mode  cfg-3" >> $out
  TOAST_HIP_PCG_SCALARS=$mode python workflows/mapmaker_pcg.py --no-filter 2>&1 | grep -i "iteration\|MapMaker\|PCG" >> $out
done
cat $out
