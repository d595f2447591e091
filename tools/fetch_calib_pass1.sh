#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03q; W=/tmp/r03q; mkdir -p $O $W
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/f -o f -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib > $O/calib.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $W/k -o k -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib >> $O/calib.log 2>&1
python3 - $W $O/calib.txt <<'PY'
import sys, glob, csv
o = open(sys.argv[2], "w")
for f in glob.glob(sys.argv[1] + "/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        o.write("%s %s %s KiB\n" % (r["Kernel_Name"][:40], r["Counter_Name"], r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/k/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        o.write("%s calls %s avg %.3f ms\n" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
cat $O/calib.txt; grep "bytes" $O/calib.log | head -2
