#!/bin/bash
# round 5, end of round: a last soak of the repeat-run stress on the final build, and the K = 31 slice of configs[4] once more
O=gpurun_out/end; mkdir -p $O
for leg in "30genes 300" "syn_pe_s0 180"; do set -- $leg
  ( timeout 700 python tools/stress_digest.py --case $1 --repeats 100000 --assemble-every 4 --seconds $2 > $O/stress_$1.log 2>&1; echo "rc=$?" >> $O/stress_$1.log ); tail -3 $O/stress_$1.log
done
timeout 700 python bench.py --config 4s --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0 > $O/bench_4s.json 2> $O/bench_4s.err
python - <<'P'
import json
j=json.loads(open("gpurun_out/end/bench_4s.json").read().strip().splitlines()[-1])
print("4s", round(j["value"]/1e6,2), "M reads/s", round(j["ms_per_step"]), "ms/step", j["config"].get("transcripts_sha256_16"))
P
