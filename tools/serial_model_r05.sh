#!/bin/bash
# round 5: the serialized N-rank model (HISTORY.md section 6: ranks share one GPU, compute one at a time, collectives through
# host memory) at a quarter of configs[2], with the components labelled on the owner shards (default) and on a replicated table
mkdir -p gpurun_out/serial
export SHN_BENCH_BACKEND=gloo
run() { # name, ranks, env
  env $3 timeout 900 python bench.py --gpus $2 --scaling strong --genes 5000 --reads 25000000 --K 25 --steps 2 --warmup 1 --no-cpu-baseline --overlap-steps 0 \
      > gpurun_out/serial/$1.json 2> gpurun_out/serial/$1.err
  echo "$1 rc=$?"
}
run n1 1 SHN_X=0
run n2 2 SHN_X=0
run n4 4 SHN_X=0
run n4_replicated 4 SHN_OWNER_LABELS=0
run n2_replicated 2 SHN_OWNER_LABELS=0
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/serial/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    print(f, "value", round(j["value"]/1e6,2), "ms/step", round(j["ms_per_step"]), j["config"].get("transcripts_sha256_16"))
    for k in ("host_stage_seconds_per_step","host_stage_seconds_per_step_slowest_rank"):
        v=j.get(k) or j["config"].get(k)
        if v: print("   ",k,{a:round(b,3) for a,b in v.items()})
P
tail -5 gpurun_out/serial/n4.err
