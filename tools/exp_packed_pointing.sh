#!/bin/bash
# PCG iteration of workflows/mapmaker_pcg.py with the solver's packed pointing cache (default) and without
# (TOAST_HIP_PACKED_POINTING=0): cfg-3, the configs[3] shard, configs[1] size.  Run on the GPU box (profiles/r03_g).
out=${1:-gpurun_out/r03p/packed.txt}
mkdir -p $(dirname $out); : > $out
run() {
  echo "== TOAST_HIP_PACKED_POINTING=$1  $2" >> $out
  shift_args="${@:3}"
  TOAST_HIP_PACKED_POINTING=$1 python workflows/mapmaker_pcg.py --no-filter $shift_args 2>&1 | grep -i "MapMaker\|pcg_iterations\|median wall\|relative residual" >> $out
}
for p in 0 1 0 1; do run $p "cfg-3"; done
for p in 0 1; do run $p "configs[3] shard" --ndet 512 --minutes 240; done
for p in 0 1 0 1; do run $p "configs[1] size" --ndet 64 --minutes 60 --rate 100 --nside 512 --iter 30; done
cat $out
