#!/bin/bash
# round 6: two batches in flight at configs[2] with the graph threads' streams at default / highest priority
mkdir -p gpurun_out/r6
for pr in 0 1 0 1; do
  SHN_FORK_PRIORITY=$pr python bench.py --steps 2 --warmup 1 --overlap-steps 6 --no-cpu-baseline 2>/dev/null > gpurun_out/r6/ov_pr$pr.json
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/ov_pr$pr.json")); o = d["overlap"]; b = o["stage_seconds_per_step_side_by_side"]
print("fork priority $pr: sequential %.0f ms, side by side %.0f ms;" % (d["ms_per_step"], o["ms_per_step"]),
      " ".join("%s %.3f" % (k, b[k]) for k in ("count", "extension", "partition+route", "graph unitigs (GPU)", "graph", "sparse flow", "post")), d["config"]["transcripts_sha256_16"])
PY
done
