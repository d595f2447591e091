#!/bin/bash
# One rocprofv3 counter pass over a command, summed per kernel: bash tools/pmc_kernels.sh "FETCH_SIZE" out_name python3 tools/ext_probe.py ...
# (counters in their own pass, never with tracing; the program follows `--` directly).  Output: gpurun_out/pmc_<out_name>.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
PMC="$1"; NAME="$2"; shift 2
W=/tmp/pmc_$NAME; rm -rf $W; mkdir -p $W gpurun_out
rocprofv3 --pmc $PMC --output-format csv -d $W -o p -- "$@" > gpurun_out/pmc_$NAME.log 2>&1
F=$(find $W -name "*counter_collection.csv" | head -1)
python3 - "$F" > gpurun_out/pmc_$NAME.txt <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(tot, key=lambda k: -max(tot[k].values())):
    print("%-60s launches %6d  " % (k[:60], cnt[k]) + "  ".join("%s total %.4g per launch %.4g" % (c, v, v / max(1, cnt[k])) for c, v in sorted(tot[k].items())))
PY
head -25 gpurun_out/pmc_$NAME.txt
