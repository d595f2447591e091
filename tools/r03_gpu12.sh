#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03l
mkdir -p $O
timeout 900 python tools/count_probe.py > $O/count_probe.log 2>&1
cat $O/count_probe.log | grep "b1="
