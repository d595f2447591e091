"""Repeat the whole single-GPU pipeline on fixed inputs and require identical products every time (the walks are a
parallel fixpoint, the contig stage runs on a host thread beside them, partitions on a thread pool).
usage: python tools/stress_pipeline.py [repeats=30]"""
import os, sys, time, hashlib
os.environ.setdefault("SHN_EXT_AUDIT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from shannon_amd import device, pipeline, kmers_for_component as kfc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
ctx = device.Context(0)


def digest(R):
    h = hashlib.sha256()
    for c in R.extension.contigs:
        h.update(c.encode())
    h.update(repr(sorted((k, sorted(v.items())) for k, v in R.extension.connections.items())).encode())
    for name, seq in R.final.items():
        h.update(name.encode()); h.update(seq.encode())
    return h.hexdigest()[:16], len(R.extension.contigs), len(R.final)


for label, args in (("configs[1]: 10M reads, one family", dict(n_pairs=5_000_000, genes=1, families=0)),
                    ("2M reads, 300 genes", dict(n_pairs=1_000_000, genes=300, families=0)),
                    ("10M reads, 4 families", dict(n_pairs=5_000_000, genes=1, families=4))):
    r1, r2 = bench.gen_reads(args["n_pairs"], 20240501, args["genes"], dev, families=args["families"])
    d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
    store = kfc.ReadStore(r1, r2)
    seen = {}
    t0 = time.time()
    for i in range(N):
        R = pipeline.assemble_resident(ctx, d1, d2, store, K=25, sample="s", seed=1)
        seen.setdefault(digest(R), []).append(i)
    print("%-36s %d runs, %.2f s each: %s" % (label, N, (time.time() - t0) / N, "IDENTICAL " + str(list(seen)[0]) if len(seen) == 1 else "DIFFERENT " + str(seen)), flush=True)
    d1.close(); d2.close()
