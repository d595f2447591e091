#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into a small text summary for profiles/.
usage: summarize_prof.py <kernel_stats.csv> [<pmc_fetch counter_collection.csv> <pmc_write counter_collection.csv>]"""
import csv, sys, re, collections


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    print("%-60s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows:
        n = short(r["Name"])
        if n.startswith(("at::", "rocprim", "hipcub")) and float(r["Percentage"]) < 1.0:
            continue
        print("%-60s %6s %12.3f %12.1f %7s" % (n[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    for label, path in zip(("FETCH_SIZE", "WRITE_SIZE"), sys.argv[2:4]):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != label:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print("\n%s per kernel (rocprofv3 --pmc %s; unit = KiB as reported; FETCH_SIZE on gfx950 reads 1/2 of wide streaming bytes -- MI355X_MICROARCH.md HBM)" % (label, label))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            print("%-60s launches=%4d  total=%14.1f KiB  per_launch=%12.1f KiB" % (k[:60], n, v, v / n))


if __name__ == "__main__":
    main()
