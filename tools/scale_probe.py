#!/usr/bin/env python3
"""Development aid: run count (+ table invariants) and the extension walks alone on a synthetic input of a given
scale, printing sizes and times -- to find where a large input (BASELINE configs[2]) breaks or spends its time.

    python tools/scale_probe.py --genes 20000 --reads 100000000 [--check-table] [--no-extend]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--K", type=int, default=25)
    ap.add_argument("--check-table", action="store_true")
    ap.add_argument("--no-extend", action="store_true")
    args = ap.parse_args()
    import bench
    from shannon_amd import device, extension_correction as ec
    dev = torch.device("cuda", 0)
    t0 = time.time()
    r1, r2 = bench.gen_reads(args.reads // 2, 20240501, args.genes, dev)
    print("gen %.1f s" % (time.time() - t0), flush=True)
    ctx = device.Context(0)
    t0 = time.time()
    sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
    ctx.sync()
    print("upload+pack %.1f s" % (time.time() - t0), flush=True)
    free, tot = torch.cuda.mem_get_info()
    print("HBM free %.1f / %.1f GB" % (free / 1e9, tot / 1e9), flush=True)
    for rep in range(2):
        t0 = time.time()
        table = device.count_k1mers(ctx, sets, args.K + 1, True)
        ctx.sync()
        print("count %.3f s: %d distinct canonical, %d windows; timers %s" % (time.time() - t0, len(table), table.total,
              {k: (round(v[0], 1), v[1]) for k, v in ctx.timers().items() if k.startswith("count") or k.startswith("table")}), flush=True)
        ctx.timer_reset()
        if rep == 0:
            table.close()
    free, tot = torch.cuda.mem_get_info()
    print("HBM free after count %.1f GB" % (free / 1e9), flush=True)
    if args.check_table:
        t0 = time.time()
        keys, cnts = table.download()
        dup = int((keys[1:] == keys[:-1]).sum())
        print("table check: adjacent duplicate keys %d, sum counts %d (windows %d), weight>=3: %d, ==1: %d  (%.1f s)"
              % (dup, int(cnts.sum(dtype=np.uint64)), table.total, int((cnts >= 3).sum()), int((cnts == 1).sum()), time.time() - t0), flush=True)
        # sample lookups: every 1000th key must be found with its own count
        samp = keys[::1000].copy()
        got = table.lookup(samp)
        print("lookup of every 1000th key: %d mismatches of %d" % (int((got != cnts[::1000]).sum()), len(samp)), flush=True)
        del keys, cnts
    if not args.no_extend:
        t0 = time.time()
        ext = ec.Extension(ctx, table, 3)
        ctx.sync()
        print("extend %.3f s: %d walks, %d rounds, %d steps; timers %s" % (time.time() - t0, ext.n_walks, ext.iterations, ext.total_steps,
              {k: (round(v[0], 1), v[1]) for k, v in ctx.timers().items() if k.startswith("ext")}), flush=True)
        ext.close()


if __name__ == "__main__":
    main()
