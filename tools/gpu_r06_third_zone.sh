#!/bin/bash
# round 6 (profiles/r06_e): which streams share an HBM zone, process by process -- N fresh processes per setting with the slab
# log and bench.py's zone diagnostic (TOAST_BENCH_ZONE_DIAG=1: TB/s of a read + write pass over 1 GB of a read stream and a
# chunk of the written timestream / the map; ~5.0 = same zone, ~5.7 = different zones).  $1 = tag, $2 = N, $3 = settings
tag=${1:-r06ah}; n=${2:-6}; which=${3:-"DEFAULT THIRD"}
mkdir -p gpurun_out/$tag
one() {  # $1 = label, rest = env assignments
  label=$1; shift
  for i in $(seq $n); do
    env TOAST_HIP_TRACE=1 TOAST_BENCH_ZONE_DIAG=${DIAG:-1} "$@" python bench.py --no-cpu-baseline --no-fft --no-operator-level --steps 10 --warmup 3 2>gpurun_out/$tag/$label$i.err > gpurun_out/$tag/$label$i.json
    python - gpurun_out/$tag/$label$i.json $label <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d['kernel_ms']; z = d.get('zone_diag') or {}
def rng(suffix):
    v = [x for kk, x in z.items() if kk.endswith(' vs ' + suffix) and not kk.startswith('tod2')]
    return '%.2f-%.2f' % (min(v), max(v)) if v else '-'
print('%s value %.2f scan %.3f bnw %.3f setup %.2f | reads vs P %s  vs Q %s  vs zmap %s | P-Q %s  P-zmap %s  Q-zmap %s | third %s' % (
    sys.argv[2], d['value'] / 1e9, k['scan'], k['bnw'], d.get('setup_s', 0), rng('tod2:P'), rng('tod2:Q'), rng('zmap'),
    z.get('tod2:P vs tod2:Q'), z.get('tod2:P vs zmap'), z.get('tod2:Q vs zmap'), z.get('slabs_third_zone')))
if 'read_GBs_vs_chunk' in z:
    print('   read arrays at GB', z['read_arrays_GB_offsets'])
    for n, v in z['read_GBs_vs_chunk'].items():
        print('   %-10s' % n, ' '.join('%.2f' % x for x in v))
if 'bnw_ms_by_map_place' in z:
    print('   bnw by map place:', ' '.join('%s/%s@%+.1f=%.3f' % (a[0][:3], a[2], a[3], a[1]) for a in z['bnw_ms_by_map_place']))
PY
    grep "vmm slab\|vmm survey\|plain slab" gpurun_out/$tag/$label$i.err | cut -c1-260
  done
}
for w in $which; do
  case $w in
    DEFAULT) one DEFAULT TOAST_HIP_ARENA_THIRD_ZONE=1 | tee gpurun_out/$tag/default.txt;;
    TWO) one TWO TOAST_HIP_ARENA_THIRD_ZONE=0 | tee gpurun_out/$tag/two.txt;;
    ENDS) one ENDS TOAST_HIP_ARENA_SURVEY_REFS=0 | tee gpurun_out/$tag/ends.txt;;
  esac
done
