"""Where in the seed order do the candidate contigs of BASELINE configs[2] lie?  (the question behind running the contig stage of the
first rank block beside the walks of the later ones)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from shannon_amd import device, extension_correction as ec
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(50_000_000, 20240501, 20000, dev, read_seed=20240503)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], 26)
ext = ec.Extension(ctx, t, 3)
ns = ext.n_walks
rank, lens = ext.accept(26, 75, 3)
print("walks", ns, "candidates", len(rank), "bases", int(lens.sum()))
for frac in (1 / 64, 1 / 32, 1 / 16, 1 / 8, 1 / 4, 1 / 2, 1.0):
    m = rank < ns * frac
    print("ranks below %.4f of the order: %8d candidates (%.1f %%), %11d bases (%.1f %%)" % (frac, int(m.sum()), 100.0 * m.mean(), int(lens[m].sum()), 100.0 * lens[m].sum() / lens.sum()))
