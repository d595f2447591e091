#!/bin/bash
# round 3, first GPU trip: K=31 slice through the whole path, the self-launching N-rank bench on one GPU (gloo), new batch generator
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fullsize_config2_gpu.py -x -q -s -m gpu > $O/fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/fullsize.log
timeout 600 python bench.py --config 4s --steps 2 --warmup 1 --no-cpu-baseline --overlap-steps 0 > $O/bench_4s.json 2> $O/bench_4s.err; echo "rc=$?" >> $O/bench_4s.err
SHN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --genes 2000 --reads 10000000 --steps 2 --warmup 1 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?" >> $O/bench_2rank.err
SHN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 1 --genes 2000 --reads 10000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_1rank.json 2> $O/bench_1rank.err; echo "rc=$?" >> $O/bench_1rank.err
tail -3 $O/fullsize.log; tail -2 $O/bench_4s.err; tail -2 $O/bench_2rank.err
