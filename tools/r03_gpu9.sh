#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
SHN_DEBUG=1 SHN_GRAPH_THREADS=1 timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --overlap-steps 0 2> $O/dbg.err > $O/dbg.json
grep "^\[mbgraph\]" $O/dbg.err | head -40 > $O/mbgraph_largest.txt
rm -f $O/dbg.err
timeout 1200 python -m pytest tests/test_midsize_gpu.py tests/test_distributed_gpu.py -x -q -m gpu -s > $O/mid.log 2>&1; echo "rc=$?" >> $O/mid.log
cat $O/mbgraph_largest.txt; tail -6 $O/mid.log
