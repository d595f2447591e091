#!/bin/bash
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_extension_gpu.py tests/test_stress_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > $O/ext.log 2>&1; echo "ext rc=$?" >> $O/ext.log
timeout 900 python tools/ext_probe.py 5000 25000000 25 > $O/probe_rec.log 2>&1; echo "rc=$?" >> $O/probe_rec.log
SHN_EXT_REFILL=1 timeout 900 python tools/ext_probe.py 5000 25000000 25 > $O/probe_rec_refill.log 2>&1; echo "rc=$?" >> $O/probe_rec_refill.log
tail -4 $O/ext.log; grep -v "lanes busy\|percentiles" $O/probe_rec.log | tail -12; grep -v "lanes busy\|percentiles" $O/probe_rec_refill.log | tail -11
