#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
W=/tmp/r03f
mkdir -p $O $W
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 -L) > $W/counters.txt 2>&1
grep -o "TCP_UTCL1[A-Z_a-z0-9]*\|TCC_EA0_[A-Za-z_0-9]*\|SQ_WAIT[A-Z_a-z0-9]*\|TCP_TCC[A-Za-z_0-9]*\|TCC_TAG[A-Za-z_0-9]*\|UTCL2[A-Za-z_0-9]*\|SQ_INSTS_VMEM[A-Z_a-z0-9]*\|SQ_THREAD[A-Z_a-z0-9]*" $W/counters.txt | sort -u > $O/counter_names.txt
P="python3 tools/ext_probe.py 5000 25000000 25"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- $P > $O/kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} $O/ ;
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $W/pmc$i -o p -- $P > $O/p$i.log 2>&1
  python3 - $W/pmc$i $O/pmc$i.txt <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
with open(sys.argv[2], "w") as o:
    for k in sorted(agg):
        if "walk" in k or "mark" in k or "records" in k or "begin" in k:
            o.write(k + " " + " ".join("%s=%.4g(n=%d)" % (c, v, cnt[(k, c)]) for c, v in sorted(agg[k].items())) + "\n")
PY
done
for f in $O/p1.log $O/p2.log $O/p3.log $O/p4.log; do tail -n 2 $f; done
cat $O/pmc*.txt | head -40
