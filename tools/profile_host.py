"""cProfile of the host side of partition+route on an F-family mixture (single GPU holding the whole job).
usage: python tools/profile_host.py [reads=40000000] [families=4]"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from shannon_amd import device, extension_correction as ec, kmers_for_component as kfc
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, 1, dev, read_seed=20240503, families=F)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], 26, True)
res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
print("contigs", len(res.contigs), "big", len(res.big_components), "remaining", len(res.remaining))
for rep in range(2):
    t0 = time.time()
    part = kfc.kmers_for_component(ctx, res, d1, d2, 25, 500, want_rows=False)
    print("kmers_for_component %.1f ms" % ((time.time() - t0) * 1e3), {n: len(v) for n, v in part["routes"].items()})
pr = cProfile.Profile()
pr.enable()
part = kfc.kmers_for_component(ctx, res, d1, d2, 25, 500, want_rows=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
