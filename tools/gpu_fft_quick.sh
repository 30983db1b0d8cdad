#!/bin/bash
# FFT tests + per-kernel times of one noise-weighting call (run on the GPU box from the repo root): $1 = tag
tag=${1:-x}
python -m pytest tests/test_gpu_fft.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fftq_$tag -o fft -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_timing.py 1024 > $GRAFT_REPO_ROOT/gpurun_out/fftq_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/fftq_$tag/*/fft_results.db 2>/dev/null | head -8 || python tools/rocpd_summary.py gpurun_out/fftq_$tag/fft_results.db | head -8
grep "^call" gpurun_out/fftq_$tag.log | tail -2
