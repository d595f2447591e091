#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_graph.sh <libA.so> <libB.so> [rounds]; the laps of the largest partition's
# graph stage and the bench line of every run land in gpurun_out/ab_*.  (SHN_HIP_LIB picks the build.)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
A=$1; B=$2; N=${3:-2}
for i in $(seq 1 $N); do
  for v in A B; do
    L=$A; [ $v = B ] && L=$B
    SHN_HIP_LIB=$PWD/$L SHN_DEBUG_PARTS=1 SHN_GRAPH_LAPS=3000000 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0 > gpurun_out/ab_${v}${i}.json 2> gpurun_out/ab_${v}${i}.err
  done
done
