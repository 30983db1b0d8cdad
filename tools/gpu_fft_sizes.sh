cd /tmp && export TMPDIR=/tmp
for ns in 360000 720000 1440000; do
nd=$((737280000/ns))
FFT_N_SAMP=$ns TOAST_HIP_FFT_POINTS=8,8,8 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fftq_n$ns -o fft -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_timing.py $nd > $GRAFT_REPO_ROOT/gpurun_out/fftq_n$ns.log 2>&1
done
