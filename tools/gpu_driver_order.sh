#!/bin/bash
# The default bench.py line as the FIRST bench process on a box after the complete GPU test suite and smoke() -- the order
# in which the driver runs things at the end of a round -- followed by two more default lines on the same box.
# $1 = tag.  Run on the GPU box.
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $out/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $out/smoke.txt
for i in 1 2 3; do
  python bench.py > $out/bench$i.json 2> $out/bench$i.err
  python tools/bench_line.py -v "bench $i=$out/bench$i.json"
done | tee $out/lines.txt
cat $out/tests.txt $out/smoke.txt
