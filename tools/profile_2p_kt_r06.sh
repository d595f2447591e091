#!/bin/bash
# round 6: kernel statistics of one step of --config 2p (graph stage on one thread and one stream, as rocprofv3 needs) + the
# per-partition time line of a normal step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
O=$PWD/gpurun_out/r6; W=/tmp/r6_2p; mkdir -p $O $W
SHN_DEBUG_PARTS=1 python3 bench.py --config 2p --steps 1 --warmup 1 --overlap-steps 0 --no-cpu-baseline > /dev/null 2> $O/2p_parts.err
grep "^\[parts\]" $O/2p_parts.err | tail -14
export SHN_GRAPH_THREADS=1 SHN_GRAPH_FORK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python3 bench.py --config 2p --no-cpu-baseline --overlap-steps 0 --steps 1 --warmup 1 > $O/kt_2p.log 2>&1
KS=$(find $W/kt -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_prof.py "$KS" > $O/r06_config2p_summary.txt 2> $O/summarize_2p.err
head -60 $O/r06_config2p_summary.txt
