#!/bin/bash
# round 6: seeds that a lower rank on a forced chain takes first are never launched (ext_chain_has_lower): parity first, then the
# hop count against the bulk walker's time at BASELINE configs[2]
mkdir -p gpurun_out/r6
python -m pytest tests/test_extension_gpu.py tests/test_midsize_gpu.py tests/test_a_stress_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r6/settle_tests.txt
cat gpurun_out/r6/settle_tests.txt
for h in ${HOPS:-0 1 2 3 4}; do
  SHN_EXT_SETTLE_HOPS=$h python bench.py --steps 4 --warmup 1 --overlap-steps 0 --no-cpu-baseline > gpurun_out/r6/settle_h$h.json 2> gpurun_out/r6/settle_h$h.err
  python - $h <<'PY'
import json, sys
h = sys.argv[1]
try:
    d = json.load(open("gpurun_out/r6/settle_h%s.json" % h))
except Exception as ex:
    print("hops", h, "FAILED", ex); print(open("gpurun_out/r6/settle_h%s.err" % h).read()[-1500:]); sys.exit(0)
k = d["kernel_ms_per_step"]; c = d["config"]
print("hops", h, "ms/step %.1f" % d["ms_per_step"], "ext %.1f" % k.get("extend", 0), "fresh %.1f" % k.get("extend.walk_fresh", 0), "thread %.1f" % k.get("extend.walk_thread", 0),
      "wave %.1f" % k.get("extend.walk_wave", 0), "begin %.1f" % k.get("extend.begin", 0), "mark %.1f" % k.get("extend.mark", 0), "rounds", c["extension_iterations"],
      "steps", c["extension_walk_steps"], "settled", c.get("extension_walks_settled_by_chain"), "sha", c["transcripts_sha256_16"], "host ext %.3f" % c["host_stage_seconds_per_step"].get("ext.gpu_walks", 0))
PY
done
