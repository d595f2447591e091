"""Time one rank's share of the sharded extension on an F-family mixture (one GPU emulating rank r of W).
usage: python tools/shard_timing.py [reads=40000000] [families=8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from shannon_amd import device, extension_correction as ec
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, 1, dev, read_seed=20240503, families=F)
ctx = device.Context(0)
sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
t = device.count_k1mers(ctx, sets, 26, True)
print("distinct k1-mers", len(t))
for W, rk in ((1, 0), (F, 0), (F, F - 1), (F, 0)):
    ctx.sync(); t0 = time.time()
    e = ec.Extension(ctx, t, 3, shard=(W, rk) if W > 1 else None)
    ctx.sync(); dt = time.time() - t0
    print("W=%d rank %d: %.1f ms, %d walks, %d rounds, %d steps" % (W, rk, dt * 1e3, e.n_walks, e.iterations, e.total_steps))
    e.close()
