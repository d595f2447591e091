#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into a small text summary (+ a traffic JSON) for profiles/.
usage: summarize_prof.py <kernel_stats.csv> [<pmc FETCH_SIZE counter_collection.csv> <pmc WRITE_SIZE counter_collection.csv> [traffic.json]]

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB as reported -> bytes): FETCH_SIZE on gfx950 tallies 128-B
requests at 64 B (MI355X_MICROARCH.md, HBM), WRITE_SIZE reads exactly.  The doubling is calibrated for wide streaming
reads; for the gather-dominated walk kernels it is an upper bound."""
import csv, sys, re, collections, json

# bench.py timer name -> kernel name prefix in the profile
TIMER_KERNEL = {"count.direct": "count_direct_kernel", "count.hist1": "hist1_kernel", "count.scatter1": "scatter1_kernel", "count.hist2": "hist_keys_kernel",
                "count.scatter2": "scatter_keys_kernel<false>", "count.buckets": "buckets_kernel<false>", "route": "route_kernel",
                # (round 5: per template instance -- the first round of a rank block and the re-run rounds are different kernels)
                "extend.walk_fresh": "ext_walk_kernel<true>", "extend.walk_thread": "ext_walk_kernel<false>", "extend.walk_wave": "ext_walk_long_kernel",
                "extend.mark": "ext_mark_kernel", "extend.begin": "ext_round_begin_kernel",
                "extend.adjacency": "ext_records_kernel",
                "count.sk_emit": "sk_scan_kernel", "count.sk_hist2": "skr_hist_kernel", "count.sk_scatter2": "skr_scatter_kernel",
                "count.sk_buckets": "sk_buckets_sorted_kernel<true, 256, 1024, 0>", "count.sk_buckets2": "sk_buckets_sorted_kernel<true, 256, 2048, 0>",
                # round 6: the contig stage's round kernels, the read -> graph mapping, the LP trials, the fixpoint audit
                "contig.hits": "cg_hits_kernel", "contig.cover": "cg_cover_kernel", "contig.compact": "cg_acc_compact_kernel",
                "graph.kp_search": "kp_search_all", "graph.kp_classify": "kp_classify", "graph.seed_scan": "seed_scan_reads_kernel",
                "graph.dd_insert": "dd_insert", "lp.trials": "lp_trials_coop_kernel", "extend.audit": "ext_audit_nodes_kernel"}


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:70]


def pmc(path, label):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != label:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    print("%-60s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows:
        n = short(r["Name"])
        if n.startswith(("at::", "rocprim", "hipcub")) and float(r["Percentage"]) < 1.0:
            continue
        print("%-60s %6s %12.3f %12.1f %7s" % (n[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    if len(sys.argv) < 4:
        return
    fetch, write = pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE")
    for label, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        print("\n%s per kernel (rocprofv3 --pmc %s, own pass; KiB as reported)" % (label, label))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("%-60s launches=%4d  total=%14.1f KiB  per_launch=%12.1f KiB" % (k[:60], n, v, v / n))
    traffic = {}
    print("\nHBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE")
    for timer, kern in TIMER_KERNEL.items():
        f = [(n, v) for k, (n, v) in fetch.items() if k.startswith(kern)]
        w = [(n, v) for k, (n, v) in write.items() if k.startswith(kern)]
        if not f or not w:
            continue
        fn, fv = sum(x[0] for x in f), sum(x[1] for x in f)
        wn, wv = sum(x[0] for x in w), sum(x[1] for x in w)
        t = 2.0 * fv / fn * 1024.0 + wv / wn * 1024.0
        traffic[timer] = t
        print("%-22s %-36s launches=%4d  fetch=%10.1f MB x2  write=%10.1f MB  traffic=%10.1f MB" % (timer, kern, fn, fv / fn * 1024 / 1e6, wv / wn * 1024 / 1e6, t / 1e6))
    if len(sys.argv) > 4:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --no-cpu-baseline --steps 1 --warmup 0",
                   "formula": "2*FETCH_SIZE + WRITE_SIZE, bytes per launch", "traffic_bytes_per_launch": traffic}, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
