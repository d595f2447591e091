import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from shannon_amd import device, extension_correction as ec
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(5_000_000, 20240501, 1, dev)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], 26, True)
ext = ec.Extension(ctx, t, 3)
nr, nl, tw = ext.stats()
live = nr != 0xFFFFFFFF
ln = 26 + nr[live].astype(np.int64) + nl[live].astype(np.int64)
print("walks", len(nr), "alive", int(live.sum()), "len>=75", int((ln >= 75).sum()), "bases of those", int(ln[ln >= 75].sum()), "max len", int(ln.max()))
T = {}
res = ec.run_correction(ctx, t, 3, 75, 500, timings=T)
print("contigs accepted", len(res.contigs), "bases", sum(len(c) for c in res.contigs), T)
