#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03q; mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -i -E "RDREQ|WRREQ|FETCH_SIZE|WRITE_SIZE|BUBBLE" $O/counters_all.txt | cut -c1-400 | head -60 > $O/counters.txt
wc -l $O/counters_all.txt
head -c 6000 $O/counters.txt
