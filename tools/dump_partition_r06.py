"""round 6: one partition of bench.py --config 2p (the one of median size) written out for host-side work on the graph surgery
without a GPU: contigs, K, the routed reads' codes.  usage (GPU box): python tools/dump_partition_r06.py gpurun_out/r6/part_2p.npz
Read back by tools/host_partition_r06.py (shn_mbgraph_run with ctx = NULL: the host-only graph stage, same surgery code)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench, shannon_amd
shannon_amd.malloc_tune()
from shannon_amd import device, pipeline, kmers_for_component as kfc
out = sys.argv[1]
K = 25
dev = torch.device("cuda", 0)
seed = 20240501
r1, r2 = bench.gen_reads(10_000_000, seed, 1, dev, read_seed=seed + 2, exon_len=(80, 600), chain_exons=30000)
ctx = device.Context(0)
sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
store = kfc.ReadStore(r1, r2)
R = pipeline.assemble_resident(ctx, sets[0], sets[1], store, K=K, sample="bench", seed=1, timings={}, keep_partitioning=True)
P = R.partitioning
names = sorted(P["routes"], key=lambda nm: len(P["routes"][nm]))
nm = names[len(names) // 2]
contigs = P["new_components"][nm]
idx = np.asarray(P["routes"][nm])
b1, o1, rc1, enc = store.gather_codes(idx, 1)
print("partition", nm, "contigs", len(contigs), "routed", len(idx), "codes", np.asarray(b1).shape, np.asarray(b1).dtype, "enc", enc, "transcripts", len(R.final))
np.savez_compressed(out, contigs=np.array(contigs, dtype=object), K=K, b1=np.asarray(b1), o1=np.asarray(o1), rc1=np.asarray(rc1), enc=enc, name=nm)
print("wrote", out, os.path.getsize(out), "bytes")
