#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $O
cd $GRAFT_REPO_ROOT
for D in 262144 2000000 16000000; do
SHN_EXT_DENSE=$D timeout 900 python3 tools/ext_probe.py 20000 100000000 25 > $O/probe_$D.txt 2> $O/probe_$D.err; echo "dense=$D rc=$?"
grep -E "^extension|extend.mark|extend.walk |extend.walk_thread|extend " $O/probe_$D.txt | head -6
done
