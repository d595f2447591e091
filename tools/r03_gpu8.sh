#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu -x --durations=8 > $O/suite.log 2>&1; echo "suite rc=$?" >> $O/suite.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --overlap-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?" >> $O/bench_c2.err
tail -12 $O/suite.log; tail -2 $O/bench_c2.err
