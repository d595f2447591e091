#!/bin/bash
# The measurement recipe behind profiles/r02_*: run on a GPU box (gpurun -- 'bash tools/profile_r02.sh'), then
#   python tools/summarize_prof.py gpurun_out/r02/kt/*/kt_kernel_stats.csv gpurun_out/r02/pmc_fetch/*/f_counter_collection.csv \
#          gpurun_out/r02/pmc_write/*/w_counter_collection.csv profiles/r02_traffic_config2.json > profiles/r02_config2_summary.txt
# Workload: the default bench (BASELINE configs[2], 100M reads / 20,000 genes).  Counters are collected in their own passes
# (never together with tracing).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp; O=gpurun_out/r02; mkdir -p $O
if [ -z "$SKIP_BENCH" ]; then python bench.py --steps 20 --warmup 5 > $O/bench_config2.json 2> $O/bench_config2.err; fi
# (the profiled passes run the graph stage on one thread and one stream: rocprofv3 aborts when threads it has not seen create streams)
export SHN_GRAPH_THREADS=1 SHN_GRAPH_FORK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --overlap-steps 0 --steps 2 --warmup 1 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/pw.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete          # keep stats + counters, drop bulky traces
find $O -name "*.csv" | head -20
tail -c 600 $O/bench_config2.json
