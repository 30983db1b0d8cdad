#!/bin/bash
# PCG iteration time with the vector updates fused into the dot products that read their output
# (toast_hip_pcg_step_dot_dev, _precond_diag_dot_dev) and with separate launches (TOAST_HIP_PCG_FUSE=0), at
# configs[1] size (64 x 360 000 @ 100 Hz, Nside 512, 30 iterations) and at cfg-3.  Run on the GPU box (profiles/r03_c).
out=${1:-gpurun_out/r03i/pcg_fuse.txt}
mkdir -p $(dirname $out); : > $out
for fuse in 0 1 0 1; do
  echo "== TOAST_HIP_PCG_FUSE=$fuse  configs[1] size" >> $out
  TOAST_HIP_PCG_FUSE=$fuse python workflows/mapmaker_pcg.py --ndet 64 --minutes 60 --rate 100 --nside 512 --iter 30 --no-filter 2>&1 | grep -i "iteration\|MapMaker\|PCG" >> $out
done
for fuse in 0 1 0 1; do
  echo "== TOAST_HIP_PCG_FUSE=$fuse  cfg-3" >> $out
  TOAST_HIP_PCG_FUSE=$fuse python workflows/mapmaker_pcg.py --no-filter 2>&1 | grep -i "iteration\|MapMaker\|PCG" >> $out
done
cat $out
