#!/bin/bash
# bench.py in fresh processes under a list of environments, one summary line each (tools/bench_line.py).
# Usage (GPU box): tools/gpu_bench_matrix.sh TAG "ENV=a ENV2=b" "ENV=c" ...     ("" = the defaults)
#   BENCH_ARGS="--no-operator-level --no-cpu-baseline"   REPS=2   TRACE=1 (TOAST_HIP_TRACE=1: slab / zone log lines into *.err)
# Round 4's runs through this: TOAST_HIP_ARENA_INTERLEAVE=1/0 in alternating processes, TOAST_HIP_ARENA_CHUNK_MB=512..4096
# (profiles/r04_a); round 5: TOAST_HIP_ARENA_STREAM_GB / _BUILDER, TOAST_HIP_COMM_PEER_WIDTH (profiles/r05_d).
tag=${1:-matrix}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
[ $# -eq 0 ] && set -- ""
for rep in $(seq 1 ${REPS:-1}); do
  i=0
  for envs in "$@"; do
    i=$((i + 1))
    f=$out/bench_${i}_$rep
    env $envs ${TRACE:+TOAST_HIP_TRACE=1} python bench.py $BENCH_ARGS > $f.json 2> $f.err
    [ -n "$TRACE" ] && grep "vmm slab" $f.err
    python tools/bench_line.py -v "[${envs:-defaults}] rep $rep=$f.json"
  done
done | tee $out/lines.txt
