#!/bin/bash
# round 5: the N-rank code path as a one-rank job at configs[2] (the whole choreography on one GPU, RCCL backend): stage times
mkdir -p gpurun_out/dist1
B="python bench.py --force-distributed --steps 2 --warmup 1 --no-cpu-baseline --overlap-steps 0"
SHN_OWNER_LABELS=2 timeout 900 $B > gpurun_out/dist1/owner.json 2> gpurun_out/dist1/owner.err; echo rc=$?
#timeout 900 $B > gpurun_out/dist1/plain.json 2> gpurun_out/dist1/plain.err; echo rc=$?
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/dist1/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    print(f, "value", round(j["value"]/1e6,2), "ms/step", round(j["ms_per_step"]), j["config"].get("transcripts_sha256_16"))
    v=j.get("host_stage_seconds_per_step") or j["config"].get("host_stage_seconds_per_step")
    print("   ",{a:round(b,3) for a,b in v.items()})
P
tail -3 gpurun_out/dist1/owner.err
