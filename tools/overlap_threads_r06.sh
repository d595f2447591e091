#!/bin/bash
# round 6: the two-batches-in-flight figure of configs[2] against the graph stage's host threads (the back half's 16 threads + the
# front half's own share one cgroup quota of 16 cpus)
mkdir -p gpurun_out/r6
for gt in 16 12 8 6; do
  SHN_GRAPH_THREADS=$gt python bench.py --steps 2 --warmup 1 --overlap-steps 6 --no-cpu-baseline 2>/dev/null > gpurun_out/r6/ov_gt$gt.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/ov_gt$gt.json")); o = d["overlap"]; b = o["stage_seconds_per_step_side_by_side"]
print("graph threads $gt: sequential %.0f ms, side by side %.0f ms;" % (d["ms_per_step"], o["ms_per_step"]),
      " ".join("%s %.3f" % (k, b[k]) for k in ("count", "extension", "partition+route", "graph unitigs (GPU)", "graph", "sparse flow", "post")))
PY
done
