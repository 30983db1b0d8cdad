for w in cfg3 cfg5g; do
TOAST_HIP_PAIR=0 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$w single', d['ms_per_step'], d['kernel_ms'])"
python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$w pair  ', d['ms_per_step'], d['kernel_ms'])"
done
