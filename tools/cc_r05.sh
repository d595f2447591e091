#!/bin/bash
# round 5: the owner-shard labelling -- its unit test, then the N-rank tests that now run through it
mkdir -p gpurun_out/cc
timeout 900 python -m pytest tests/test_cc_shards_gpu.py -x -q -m gpu > gpurun_out/cc/unit.txt 2>&1
tail -25 gpurun_out/cc/unit.txt
timeout 1500 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/cc/dist.txt 2>&1
tail -40 gpurun_out/cc/dist.txt
