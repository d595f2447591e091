#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_extension_gpu.py tests/test_distributed_gpu.py tests/test_midsize_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
export SHN_BENCH_BACKEND=gloo
timeout 600 python bench.py --gpus 1 --force-distributed --genes 5000 --reads 25000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/n1d.json 2> $O/n1d.err
timeout 900 python bench.py --gpus 2 --genes 5000 --reads 25000000 --steps 2 --warmup 1 > $O/n2.json 2> $O/n2.err
timeout 1200 python bench.py --gpus 4 --genes 5000 --reads 25000000 --steps 2 --warmup 1 > $O/n4.json 2> $O/n4.err
tail -3 $O/tests.log
for f in n1d n2 n4; do echo "== $f"; python tools/print_stages.py $O/$f.json 2>&1 | grep -v "route\.\|ext\.filter\|ext\.emit\|ext\.comp" | head -40; done
