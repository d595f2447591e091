"""round 6: how many of the walkers' steps end up in the final claims?  BASELINE configs[2] (or argv: genes reads): the steps of the
walks that survive (= the k1-mers finally claimed) against the steps executed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from shannon_amd import device, extension_correction as ec
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, dev)
ctx = device.Context(0)
sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
t = device.count_k1mers(ctx, sets, 26, both_strands=True)
ext = ec.Extension(ctx, t, 3)
rank, nr, nl, tw = ext.live_stats(0)
steps_alive = int(nr.astype(np.int64).sum() + nl.astype(np.int64).sum())
print("k1-mers %d (oriented %d), seeds %d, alive walks %d, k1-mers finally claimed %d (%.1f %% of the oriented), steps executed %d (fresh %d) = %.2f x the claimed; rounds %d"
      % (len(t), 2 * len(t), ext.n_walks, len(rank), steps_alive + len(rank), 100.0 * (steps_alive + len(rank)) / (2 * len(t)), ext.total_steps, ext.fresh_steps,
         ext.total_steps / max(1, steps_alive), ext.iterations))
L = nr.astype(np.int64) + nl.astype(np.int64) + 1
for q in (50, 90, 99, 99.9, 100):
    print("  alive walk length percentile %5.1f: %d" % (q, int(np.percentile(L, q))))
print("  alive walks of >= 75 k1-mers: %d holding %d k1-mers" % (int((L >= 75).sum()), int(L[L >= 75].sum())))
