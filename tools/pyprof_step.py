#!/usr/bin/env python3
"""Development aid: cProfile of one pipeline step at the default workload (where does the interpreter spend time between the
native calls?).  python tools/pyprof_step.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from shannon_amd import device, pipeline, kmers_for_component as kfc

dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(50_000_000, 20240501, 20000, dev, read_seed=20240503)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
store = kfc.ReadStore(r1, r2)
run = lambda: pipeline.assemble_resident(ctx, d1, d2, store, K=25, sample="bench", seed=1)
run()
pr = cProfile.Profile()
pr.enable()
R = run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
