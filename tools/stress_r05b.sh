#!/bin/bash
# round 5, second soak: the final allocator (reuse ordered by stream-wait events), both inputs of tests/test_a_stress_gpu.py
mkdir -p gpurun_out/stress2
for leg in "30genes 420" "syn_pe_s0 240"; do set -- $leg
  ( timeout 900 python tools/stress_digest.py --case $1 --repeats 100000 --assemble-every 4 --seconds $2 > gpurun_out/stress2/$1.log 2>&1; echo "rc=$?" >> gpurun_out/stress2/$1.log ); tail -3 gpurun_out/stress2/$1.log
done
python - <<'E2'
import sys; sys.path.insert(0, ".")
from shannon_amd import _lib
E2
