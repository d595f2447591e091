#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_count_gpu.py tests/test_fullsize_gpu.py tests/test_midsize_gpu.py tests/test_e2e_gpu.py tests/test_routing_gpu.py tests/test_seeds_gpu.py -x -q -m gpu > $O/count.log 2>&1; echo "rc=$?" >> $O/count.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err
tail -4 $O/count.log
