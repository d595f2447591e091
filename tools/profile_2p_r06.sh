#!/bin/bash
# round 6: where the step of --config 2p goes (401 gpmetis partitions of repeat-linked genes): the bench line, then one step with
# the laps of EVERY partition's graph stage summed by phase (tools/sum_laps.py)
mkdir -p gpurun_out/r6
python bench.py --config 2p --steps 2 --warmup 1 --overlap-steps 0 --no-cpu-baseline > gpurun_out/r6/2p_${TAG:-base}.json 2> gpurun_out/r6/2p_${TAG:-base}.err
python - <<PY
import json
d = json.load(open("gpurun_out/r6/2p_${TAG:-base}.json")); c = d["config"]
print("2p ms/step %.0f" % d["ms_per_step"], "sha", c["transcripts_sha256_16"], {k: round(v, 3) for k, v in c["host_stage_seconds_per_step"].items() if v > 0.05})
print({k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if v > 20}, "lp calls", c["lp_calls"], "trials", c["lp_trials"])
PY
if [ -z "$NOLAPS" ]; then
SHN_GRAPH_LAPS=0 python bench.py --config 2p --steps 1 --warmup 1 --overlap-steps 0 --no-cpu-baseline > /dev/null 2> gpurun_out/r6/2p_laps_${TAG:-base}.err
python tools/sum_laps.py gpurun_out/r6/2p_laps_${TAG:-base}.err | head -40 | tee gpurun_out/r6/2p_laps_${TAG:-base}.txt
grep "bridge_all:" gpurun_out/r6/2p_laps_${TAG:-base}.err | head -5
fi
