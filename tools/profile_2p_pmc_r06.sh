#!/bin/bash
# round 6: PMC traffic per kernel at --config 2p (one step per pass; the graph stage on one thread and one stream: rocprofv3 aborts
# when threads it has not seen create streams).  Counters in their own passes, the program directly after `--`.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r06p; W=/tmp/r06p_2p; mkdir -p $O $W
export SHN_GRAPH_THREADS=1 SHN_GRAPH_FORK=0
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python3 bench.py --config 2p --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/kt1_2p.log 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc_fetch -o f -- python3 bench.py --config 2p --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/pf_2p.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc_write -o w -- python3 bench.py --config 2p --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 0 > $O/pw_2p.log 2>&1
KS=$(find $W/kt -name "*kernel_stats.csv" | head -1); FC=$(find $W/pmc_fetch -name "*counter_collection.csv" | head -1); WC=$(find $W/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/summarize_prof.py "$KS" "$FC" "$WC" $O/r06_traffic_config2p.json > $O/r06_config2p_pmc_summary.txt 2> $O/summarize_2p_pmc.err
tail -25 $O/r06_config2p_pmc_summary.txt; tail -3 $O/summarize_2p_pmc.err
