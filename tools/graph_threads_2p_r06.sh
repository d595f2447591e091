#!/bin/bash
# round 6: --config 2p against the graph stage's partition threads (the box grants 16 cpus by quota)
mkdir -p gpurun_out/r6
for rep in 1 2; do
for gt in 16 12 14 20 24; do
  SHN_GRAPH_THREADS=$gt timeout 300 python bench.py --config 2p --steps 3 --warmup 1 --overlap-steps 0 --no-cpu-baseline > gpurun_out/r6/gt2p_${gt}_$rep.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/gt2p_${gt}_$rep.json")); c = d["config"]["host_stage_seconds_per_step"]
print("rep $rep graph threads $gt: %.0f ms/step" % d["ms_per_step"], d["config"]["transcripts_sha256_16"], "graph %.2f sflow %.2f post %.2f" % (c["graph"], c["sparse flow"], c["post"]))
PY
done
done
