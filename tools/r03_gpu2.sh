#!/bin/bash
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lp_gpu.py -x -q -m gpu > $O/lp.log 2>&1; echo "lp rc=$?" >> $O/lp.log
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_lp_gpu.py --durations=15 > $O/suite.log 2>&1; echo "suite rc=$?" >> $O/suite.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?" >> $O/bench_c2.err
tail -4 $O/lp.log; tail -6 $O/suite.log; tail -2 $O/bench_c2.err
