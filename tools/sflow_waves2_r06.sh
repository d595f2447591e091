#!/bin/bash
mkdir -p gpurun_out/r6
for rep in 1 2; do
for cfg in "0 3" "40 4" "32 5" "50 4" "40 5"; do
  set -- $cfg
  SHN_SFLOW_WAVE=$1 SHN_SFLOW_WAVES_MAX=$2 timeout 300 python bench.py --config 2p --steps 3 --warmup 1 --overlap-steps 0 --no-cpu-baseline > gpurun_out/r6/wv2_$1_$2_$rep.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/wv2_$1_$2_$rep.json")); c = d["config"]["host_stage_seconds_per_step"]
print("rep $rep wave $1 max $2: %.0f ms/step" % d["ms_per_step"], d["config"]["transcripts_sha256_16"], "graph %.2f sflow %.2f post %.2f" % (c["graph"], c["sparse flow"], c["post"]))
PY
done
done
