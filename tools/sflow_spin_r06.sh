#!/bin/bash
# round 6: the sparse-flow pool's spin between the phases of a round at --config 2p, with the cgroup's throttling counters beside it
mkdir -p gpurun_out/r6
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for sp in 0 20 400 1500; do
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  SHN_SFLOW_SPIN_US=$sp SHN_DEBUG_PARTS=1 python bench.py --config 2p --steps 3 --warmup 1 --overlap-steps 0 --no-cpu-baseline > gpurun_out/r6/spin_$sp.json 2> gpurun_out/r6/spin_$sp.err
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  python - <<PY
import json
d = json.load(open("gpurun_out/r6/spin_$sp.json")); c = d["config"]["host_stage_seconds_per_step"]
print("spin $sp us: %.0f ms/step" % d["ms_per_step"], d["config"]["transcripts_sha256_16"], "graph %.2f sflow %.2f post %.2f" % (c["graph"], c["sparse flow"], c["post"]))
print("   cpu.stat before: $a")
print("   cpu.stat after : $b")
PY
  grep "stage wall" gpurun_out/r6/spin_$sp.err | tr '\n' ';'; echo
done
