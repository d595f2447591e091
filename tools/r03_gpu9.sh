#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_lp_gpu.py tests/test_midsize_gpu.py tests/test_distributed_gpu.py tests/test_e2e_gpu.py tests/test_random_parity_gpu.py -x -q -m gpu -s > $O/mid.log 2>&1; echo "rc=$?" >> $O/mid.log
SHN_DEBUG=1 SHN_GRAPH_THREADS=1 timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --overlap-steps 0 2> $O/dbg.err > $O/dbg.json
grep "^\[mbgraph\]" $O/dbg.err | head -16 > $O/mbgraph_largest.txt
rm -f $O/dbg.err
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err
cat $O/mbgraph_largest.txt; grep "passed\|failed\|K=" $O/mid.log | tail -6
