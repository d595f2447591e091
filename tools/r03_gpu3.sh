#!/bin/bash
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_lp_gpu.py tests/test_e2e_gpu.py tests/test_random_parity_gpu.py tests/test_reference_api_gpu.py -x -q -m gpu > $O/lp.log 2>&1; echo "lp rc=$?" >> $O/lp.log
timeout 900 python tools/ext_probe.py 5000 25000000 25 > $O/probe.log 2>&1; echo "rc=$?" >> $O/probe.log
tail -4 $O/lp.log; tail -30 $O/probe.log
