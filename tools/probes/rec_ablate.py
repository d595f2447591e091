"""timing of the LDS records kernel with parts switched off (SHN_REC_ABLATE; wrong records, timing only): where a trip's time goes"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from shannon_amd import device, _lib
import ctypes as C
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(50_000_000, 20240501, 20000, dev, read_seed=20240503)
ctx = device.Context(0)
sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
t = device.count_k1mers(ctx, sets, 26, True)
print("k1-mers", len(t))
from shannon_amd import extension_correction as ec
for ab in ([0, 1, 2, 3] if os.environ.get("SHN_REC_LDS") == "1" else [0]):
    os.environ["SHN_REC_ABLATE"] = str(ab)
    for rep in range(2):
        ctx.timer_reset()
        e = ec.Extension(ctx, t, 0x7FFFFFF0)            # no seed is that heavy: the records are built, nothing walks
        ctx.sync()
        tm = ctx.timers()
        e.close()
    print("SHN_REC_LDS=%s ablate=%d: adjacency %.1f ms" % (os.environ.get("SHN_REC_LDS", "0"), ab, tm["extend.adjacency"][0]))
