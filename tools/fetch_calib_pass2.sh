#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03q; W=/tmp/r03q2; mkdir -p $O $W
cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $W/a -o a -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib > $O/calib2.log 2>&1
rocprofv3 --pmc TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $W/b -o b -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib >> $O/calib2.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum WRITE_SIZE --output-format csv -d $W/c -o c -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib >> $O/calib2.log 2>&1
python3 - $W $O/calib2.txt <<'PY'
import sys, glob, csv
o = open(sys.argv[2], "w")
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        o.write("%s %s %s\n" % (r["Kernel_Name"][:12], r["Counter_Name"], r["Counter_Value"]))
PY
cat $O/calib2.txt; tail -5 $O/calib2.log
