#!/bin/bash
# The measurement recipe behind profiles/r06_*: run on a GPU box (gpurun -- 'bash tools/profile_r05.sh [config]'); the rocprofv3
# output stays in /tmp on the box, the summaries come back under gpurun_out/r06p/ and are copied to profiles/ by hand.
# Workload: the default bench (BASELINE configs[2], 100M reads / 20,000 genes) or `4s`.  Counters are collected in their own passes
# (never together with tracing); the program follows `--` directly.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
CFG=${1:-2}
O=$PWD/gpurun_out/r06p; W=/tmp/r06p_$CFG; mkdir -p $O $W
if [ -z "$SKIP_BENCH" ]; then python3 bench.py --config $CFG --steps ${STEPS:-20} --warmup 5 > $O/bench_config$CFG.json 2> $O/bench_config$CFG.err; fi
# (the profiled passes run the graph stage on one thread and one stream: rocprofv3 aborts when threads it has not seen create streams)
export SHN_GRAPH_THREADS=1 SHN_GRAPH_FORK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python3 bench.py --config $CFG --no-cpu-baseline --overlap-steps 0 --steps 3 --warmup 1 > $O/kt_$CFG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc_fetch -o f -- python3 bench.py --config $CFG --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/pf_$CFG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc_write -o w -- python3 bench.py --config $CFG --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/pw_$CFG.log 2>&1
KS=$(find $W/kt -name "*kernel_stats.csv" | head -1); FC=$(find $W/pmc_fetch -name "*counter_collection.csv" | head -1); WC=$(find $W/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/summarize_prof.py "$KS" "$FC" "$WC" $O/r06_traffic_config$CFG.json > $O/r06_config${CFG}_summary.txt 2> $O/summarize_$CFG.err
tail -c 400 $O/bench_config$CFG.json; head -30 $O/r06_config${CFG}_summary.txt
