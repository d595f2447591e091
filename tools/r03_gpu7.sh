#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03g
mkdir -p $O
SHN_EXT_XTIME=1 timeout 600 python tools/ext_probe.py 5000 25000000 25 2>&1 | grep "XTIME" | grep "round 1:\|round 9:" | tail -2 > $O/x0.log
SHN_EXT_XPLAIN=1 SHN_EXT_XTIME=1 timeout 600 python tools/ext_probe.py 5000 25000000 25 2>&1 | grep "XTIME" | grep "round 1:\|round 9:" | tail -2 > $O/x1.log
SHN_EXT_REFILL=1 SHN_EXT_XPLAIN=1 SHN_EXT_XTIME=1 timeout 600 python tools/ext_probe.py 5000 25000000 25 2>&1 | grep "XTIME" | grep "round 1:\|round 9:" | tail -2 > $O/x2.log
cat $O/x0.log $O/x1.log $O/x2.log | cut -c1-220
