#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_post_gpu.py tests/test_e2e_gpu.py tests/test_random_parity_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
SHN_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap-steps 0 2> $O/dbg.err > $O/bench_c2.json
grep "^\[post\]\|^\[find_reps\]" $O/dbg.err | tail -4 > $O/post_lines.txt; rm -f $O/dbg.err
tail -5 $O/tests.log; cat $O/post_lines.txt
