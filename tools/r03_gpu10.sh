#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_extension_gpu.py tests/test_midsize_gpu.py -x -q -m gpu > $O/ext.log 2>&1; echo "ext rc=$?" >> $O/ext.log
timeout 900 python tools/ext_probe.py 5000 25000000 25 2>&1 | grep "extension \|extend" > $O/probe.log
tail -3 $O/ext.log; cat $O/probe.log
