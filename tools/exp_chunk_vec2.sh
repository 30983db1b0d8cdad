#!/bin/bash
# Chunk size (samples per workgroup) x lane width at cfg-3: kernel times of build_noise_weighted / scan_map /
# read-write stream, one bench process per chunk size (profiles/r03_a).  Run on the GPU box.
out=${1:-gpurun_out/r03a/chunk_sweep.txt}
mkdir -p $(dirname $out); : > $out
for c in 512 1024 2048 4096; do
  TOAST_HIP_CHUNK=$c python bench.py --no-cpu-baseline --no-fft --no-operator-level --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
l=d['roofline']['lane_width']
print('chunk $c  step %.3f ms  value %.2f G/s' % (d['ms_per_step'], d['value']/1e9))
for k,v in l.items(): print('   %-22s bnw %.3f ms  scan %.3f ms  rw %.3f ms' % (k, v['build_noise_weighted_ms'], v['scan_map_ms'], v['read_write_stream_ms']))
" >> $out
done
cat $out
