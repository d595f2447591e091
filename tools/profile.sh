#!/bin/bash
# The measurement recipe behind profiles/: run on a GPU box (e.g. gpurun -- 'bash tools/profile.sh'), then
#   python tools/summarize_prof.py gpurun_out/prof/kt/kt_kernel_stats.csv gpurun_out/prof/pmc_fetch/f_counter_collection.csv \
#          gpurun_out/prof/pmc_write/w_counter_collection.csv profiles/rNN_traffic.json > profiles/rNN_final_10M_summary.txt
# Counters are collected in their own passes (never together with tracing).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp; O=gpurun_out/r1f; mkdir -p $O
python bench.py --steps 5 --warmup 1 > $O/bench_final.json 2> $O/bench_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pw.log 2>&1
find $O -name "*.csv" | head -20
find $O -name "*kernel_trace.csv" -size +30M -delete          # keep stats + counters, drop bulky traces
tail -c 400 $O/bench_final.json
