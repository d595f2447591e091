"""Parity of the sharded contig stages at benchmark scale: F families, `reads` reads, W virtual ranks on one GPU.
usage: python tools/check_sharded_scale.py [reads=40000000] [families=4] [world=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from shannon_amd import device, extension_correction as ec
from test_extension_gpu import _run_virtual_ranks
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, 1, dev, read_seed=20240503, families=F)
ctx = device.Context(0)
t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)], 26, True)
ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
print("global: contigs", len(ref.contigs), "single", len(ref.single_contigs), "remaining", [len(x) for x in ref.remaining], "big", len(ref.big_components))
results, history = _run_virtual_ranks(W, lambda rank, g: ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, shard=(W, rank), gather=g))
print("guard verdict (0 = shards are independent):", history[1])
res = results[0]
print("sharded: contigs", len(res.contigs), "single", len(res.single_contigs), "remaining", [len(x) for x in res.remaining], "big", len(res.big_components))
print("contigs equal:", res.contigs == ref.contigs, " connections equal:", res.connections == ref.connections, " components equal:", res.components == ref.components)
if res.contigs != ref.contigs:
    a, b = set(ref.contigs), set(res.contigs)
    print("only global", len(a - b), "only sharded", len(b - a))
