import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from shannon_amd import device, pipeline, kmers_for_component as kfc, extension_correction as ec
dev = torch.device("cuda", 0)
F = 8
r1, r2 = bench.gen_reads(5_000_000 * F, 20240501, 1, dev, families=F)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], 26, True)
res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
kfc.kmers_for_component(ctx, res, d1, d2, 25, 500, want_rows=False)
pr = cProfile.Profile()
pr.enable()
kfc.kmers_for_component(ctx, res, d1, d2, 25, 500, want_rows=False)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
