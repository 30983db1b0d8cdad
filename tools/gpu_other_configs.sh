#!/bin/bash
# round 6: the other BASELINE configurations at one GPU's share with the round's code (wall times of the workflows).  $1 = tag
tag=${1:-r06p}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python workflows/mapmaker_pcg.py --ndet 512 --minutes 240 > $out/cfg3shard.log 2>&1
tail -16 $out/cfg3shard.log
python workflows/ground_filter_mapmaker.py --split > $out/cfg4.log 2>&1
tail -14 $out/cfg4.log
python workflows/mapmaker_pcg.py --uncached > $out/uncached.log 2>&1
tail -8 $out/uncached.log
for w in cfg4 cfg2 cfg5g; do
  python bench.py --workload $w --no-cpu-baseline --no-operator-level --no-fft 2>/dev/null | python tools/bench_line.py /dev/stdin 2>/dev/null | head -2
done
