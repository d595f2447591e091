#!/bin/bash
# round 6: --config 2p against the size of a sparse-flow wave and the waves in flight beside the graph stage
mkdir -p gpurun_out/r6
for cfg in "0 3" "40 4" "40 6" "25 6" "100 3"; do
  set -- $cfg
  SHN_SFLOW_WAVE=$1 SHN_SFLOW_WAVES_MAX=$2 SHN_DEBUG_PARTS=1 timeout 300 python bench.py --config 2p --steps 3 --warmup 1 --overlap-steps 0 --no-cpu-baseline > gpurun_out/r6/wv_$1_$2.json 2> gpurun_out/r6/wv_$1_$2.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/wv_$1_$2.json")); c = d["config"]["host_stage_seconds_per_step"]
print("wave $1 (0 = a sixth), at most $2 in flight: %.0f ms/step" % d["ms_per_step"], d["config"]["transcripts_sha256_16"], "graph %.2f sflow %.2f post %.2f" % (c["graph"], c["sparse flow"], c["post"]))
PY
  grep "stage wall" gpurun_out/r6/wv_$1_$2.err | tr '\n' ';' | cut -c1-400; echo
done
