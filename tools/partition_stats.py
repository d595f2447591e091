#!/usr/bin/env python3
"""Per-partition figures of a full-size run (bench.py's batch): routed pairs, K-mer nodes, read cutoff, bridging steps, known /
mate paths -- what tests/test_fullsize_config2_gpu.py's choice of partitions for the oracle comparison was sized on.
    python tools/partition_stats.py [config] [n_pairs] > gpurun_out/partition_stats_<config>.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from shannon_amd import device, pipeline, kmers_for_component as kfc

cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
P = bench.PRESETS[cfg]
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else P["reads"] // 2
K = P["K"]
r1, r2 = bench.gen_reads(n_pairs, 20240501, P["genes"], torch.device("cuda", 0), read_seed=20240503, exon_len=P["exon_len"])
torch.cuda.empty_cache()
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
store = kfc.ReadStore(r1, r2)
T = {}
t = time.time()
R = pipeline.assemble_resident(ctx, d1, d2, store, K=K, sample="bench", seed=1, keep_partitioning=True, timings=T)
dt = time.time() - t
out = {"config": cfg, "n_pairs": n_pairs, "seconds": dt, "timings": {k: round(v, 3) for k, v in T.items() if isinstance(v, float)}, "partitions": []}
part = R.partitioning
for nm, rec in R.partitions.items():
    log = rec["log"]
    out["partitions"].append({"name": nm, "routed": int(len(part["routes"][nm])), "contigs": len(part["new_components"][nm]),
                              "k1mer_rows": int(part["n_k1mer_rows"][nm]), "final_nodes": log.get("final_nodes"), "nodes_after": log.get("nodes_after"),
                              "known_paths": log.get("known_paths"), "mate_paths": log.get("mate_paths"), "bridged": log.get("bridged"),
                              "components": len(rec["components"]), "singles": len(rec["singles"])})
print(json.dumps(out))
