#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_graph.sh <libA.so> <libB.so> [rounds]; the bench line of every run (and the
# graph-stage laps of the largest partition) land in gpurun_out/ab_*.  (SHN_HIP_LIB picks the build; tools/build_ab_base.sh makes A.)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for v in A B; do
    L=$A; [ $v = B ] && L=$B
    SHN_HIP_LIB=$PWD/$L SHN_DEBUG_PARTS=1 SHN_GRAPH_LAPS=3000000 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --overlap-steps 0 > gpurun_out/ab_${v}${i}.json 2> gpurun_out/ab_${v}${i}.err
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_[AB]*.json')):
    try:
        d=json.load(open(f)); h=d['config']['host_stage_seconds_per_step']
        print(f[-8:-5], round(d['ms_per_step'],1), d.get('transcripts_sha256_16'), {k:round(v,3) for k,v in h.items() if k in ('count','extension','partition+route','graph','sparse flow','post')})
    except Exception as e: print(f, 'failed', e)
PY
