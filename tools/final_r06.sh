#!/bin/bash
# round 6: the committed measurements (profiles/r06_*): PMC traffic + kernel statistics of configs[2] first (bench.py reads the newest
# traffic file of its config), then the bench lines of configs[2] (20 steps, CPU baseline, two batches in flight), 4s and 2p, and
# the kernel statistics of 2p.  gpurun -- 'bash tools/final_r06.sh'; the files come back under gpurun_out/r06p/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r06p; mkdir -p $O
SKIP_BENCH=1 bash tools/profile_r06.sh 2 > $O/profile_2.log 2>&1
cp $O/r06_traffic_config2.json profiles/r06_traffic_config2.json
python3 bench.py --config 2 --steps 20 --warmup 5 > $O/r06_bench_config2.json 2> $O/bench2.err
python3 bench.py --config 4s --steps 10 --warmup 3 --overlap-steps 0 --no-cpu-baseline > $O/r06_bench_config4s.json 2> $O/bench4s.err
python3 bench.py --config 2p --steps 5 --warmup 2 --overlap-steps 0 --no-cpu-baseline > $O/r06_bench_config2p.json 2> $O/bench2p.err
W=/tmp/r06_2p; mkdir -p $W
SHN_GRAPH_THREADS=1 SHN_GRAPH_FORK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python3 bench.py --config 2p --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 1 > $O/kt_2p.log 2>&1
KS=$(find $W/kt -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_prof.py "$KS" > $O/r06_config2p_summary.txt 2> $O/summarize_2p.err
python3 - <<PY
import json
for c in ("2", "4s", "2p"):
    try:
        d = json.load(open("$O/r06_bench_config%s.json" % c))
        r = d["roofline"]
        print(c, round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 1), "ms", d["config"]["transcripts_sha256_16"], d["config"]["steps_checked"]["all_equal"],
              "dominant", r["kernel"], round(r["frac"], 4), "traffic", r.get("traffic"), (d.get("overlap") or {}).get("ms_per_step"))
    except Exception as ex:
        print(c, "FAILED", ex)
PY
head -12 $O/r06_config2_summary.txt
