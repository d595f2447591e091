#!/bin/bash
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_extension_gpu.py tests/test_stress_gpu.py -x -q -m gpu > $O/ext.log 2>&1; echo "ext rc=$?" >> $O/ext.log
SHN_DEBUG=1 timeout 900 python tools/ext_probe.py 5000 25000000 25 > $O/probe_refill.log 2>&1; echo "rc=$?" >> $O/probe_refill.log
SHN_EXT_REFILL=0 SHN_DEBUG=1 timeout 900 python tools/ext_probe.py 5000 25000000 25 > $O/probe_old.log 2>&1; echo "rc=$?" >> $O/probe_old.log
tail -4 $O/ext.log; grep -v "shn_extend\] round\|memo_follow\|contig" $O/probe_refill.log | tail -22; grep -v "shn_extend\] round\|memo_follow\|contig" $O/probe_old.log | tail -14
