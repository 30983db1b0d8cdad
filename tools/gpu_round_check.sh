#!/bin/bash
# full GPU check of the tree: tests, smoke, default bench line (run on the GPU box from the repo root): $1 = tag
tag=${1:-x}
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_gputests.log; tail -1 gpurun_out/${tag}_smoke.log; tail -c 1500 gpurun_out/${tag}_bench.json
