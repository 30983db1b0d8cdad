# scan_map / build_noise_weighted under the two workgroup orders and chunk sizes (TLB reach vs map locality)
for c in 1024 4096 16384; do
for dm in 0 1; do
TOAST_HIP_PAIR=0 TOAST_HIP_CHUNK=$c TOAST_HIP_DET_MAJOR=$dm python bench.py --no-cpu-baseline --steps 5 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('chunk $c det_major $dm', d['ms_per_step'], d['kernel_ms'])"
done; done
