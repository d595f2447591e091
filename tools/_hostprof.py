import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from shannon_amd import device, pipeline, kmers_for_component as kfc
dev = torch.device("cuda", 0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r1, r2 = bench.gen_reads(5_000_000 * F, 20240501, 1, dev, families=F)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
store = kfc.ReadStore(r1, r2)
pipeline.assemble_resident(ctx, d1, d2, store, K=25, sample="bench", seed=1)
pr = cProfile.Profile()
pr.enable()
T = {}
pipeline.assemble_resident(ctx, d1, d2, store, K=25, sample="bench", seed=1, timings=T)
pr.disable()
print(T)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
