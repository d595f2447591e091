"""round 6: the host-only graph stage (shn_mbgraph_run, ctx = NULL) over a partition written by tools/dump_partition_r06.py --
the graph surgery's code without a GPU.  usage: python tools/host_partition_r06.py gpurun_out/r6/part_2p.npz [repeats]
Prints the wall time per run, a digest of the exported graph (nodes, edges, paths: must not move) and, with SHN_GRAPH_LAPS=0, the laps."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from shannon_amd import mbgraph_native, kmers_for_component as kfc
z = np.load(sys.argv[1], allow_pickle=True)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
contigs, K = [str(c) for c in z["contigs"]], int(z["K"])
b1, o1, rc1, enc = z["b1"], z["o1"], z["rc1"], int(z["enc"])
rows = kfc._rows_bytes(contigs, K + 1)
n_rows = len(rows) // (K + 1)
rc2 = (1 - rc1).astype(np.uint8)
best = 1e9
for it in range(reps):
    t = time.time()
    g = mbgraph_native.run_partition_handle(rows, n_rows, K, b1, o1, b1, o1, ctx=None, enc=enc, rc1=rc1, rc2=rc2)
    dt = time.time() - t
    best = min(best, dt)
    h = hashlib.sha256()
    h.update(repr(g.tables()).encode())
    print("run %d: %.3f s  digest %s" % (it, dt, h.hexdigest()[:16]), flush=True)
    g.close()
print("best %.3f s (%d contigs, %d k1-mer rows, %d routed pairs)" % (best, len(contigs), n_rows, len(rc1)))
