#!/bin/bash
# A/B: MLP-tuned IQU kernels vs the simple ones, plus parity.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('tuned rep$rep', round(j['value']/1e9,2), j['kernel_ms'])"
  TOAST_HIP_SIMPLE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('simple rep$rep', round(j['value']/1e9,2), j['kernel_ms'])"
done
