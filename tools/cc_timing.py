"""Time the pieces of the owner-shard labelling on one GPU playing rank r of W (round 5).
usage: python tools/cc_timing.py [reads=25000000] [genes=5000] [W=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from shannon_amd import device
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, dev, read_seed=20240503)
ctx = device.Context(0)
sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
t = device.count_k1mers(ctx, sets, 26, True)
n = len(t)
print("distinct k1-mers", n)
dk = torch.empty(n, dtype=torch.int64, device=dev); dc = torch.empty(n, dtype=torch.int32, device=dev)


def lap(name, t0):
    ctx.sync(); torch.cuda.synchronize()
    print("  %-40s %7.1f ms" % (name, (time.time() - t0) * 1e3))
    return time.time()


for rep in range(3):
    print("rep", rep)
    t0 = time.time()
    per = t.shard_by_minimizer(W, dk.data_ptr(), dc.data_ptr()).astype(np.int64)
    t0 = lap("shard by minimizer (whole table)", t0)
    print("  shards", per.tolist())
    off = np.concatenate([[0], np.cumsum(per)])
    r = 0
    sh = device.Table.from_pairs(ctx, dk[off[r]:].data_ptr(), dc[off[r]:].data_ptr(), int(per[r]), 26, True)
    t0 = lap("from_pairs (shard 0)", t0)
    cc = device.ComponentShards(ctx, sh, W, r)
    t0 = lap("cc create (local labels + query count)", t0)
    pq = cc.query_counts().astype(np.int64)
    qk = torch.empty(max(int(pq.sum()), 1), dtype=torch.int64, device=dev); ql = torch.empty(max(int(pq.sum()), 1), dtype=torch.int32, device=dev)
    t0 = lap("alloc queries", t0)
    cc.queries(qk.data_ptr(), ql.data_ptr())
    t0 = lap("queries", t0)
    print("  queries per k1-mer of the shard: %.3f" % (pq.sum() / max(1, per[r])), pq.tolist())
    # rank 1 answers rank 0's queries
    sh1 = device.Table.from_pairs(ctx, dk[off[1]:].data_ptr(), dc[off[1]:].data_ptr(), int(per[1]), 26, True)
    cc1 = device.ComponentShards(ctx, sh1, W, 1)
    t0 = lap("(rank 1: table + cc)", t0)
    o = np.concatenate([[0], np.cumsum(pq)])
    rcl = [int(pq[1])] + [0] * (W - 1)
    edges = torch.empty(2 * max(rcl[0], 1), dtype=torch.int64, device=dev)
    ne = cc1.answer(qk[o[1]:].data_ptr(), ql[o[1]:].data_ptr(), rcl, [0, int(per[0])] + [0] * (W - 2), edges.data_ptr())
    t0 = lap("answer (%d queries -> %d edges)" % (rcl[0], ne), t0)
    E = ne
    nodes = torch.empty(2 * max(E, 1), dtype=torch.int64, device=dev); labels = torch.empty(2 * max(E, 1), dtype=torch.int64, device=dev)
    nn = device.ComponentShards.solve(ctx, edges.data_ptr(), E, n + 1, nodes.data_ptr(), labels.data_ptr())
    t0 = lap("solve (%d edges, %d nodes)" % (E, nn), t0)
    gl = torch.empty(int(per[0]), dtype=torch.int64, device=dev)
    cc.labels(0, nodes.data_ptr(), labels.data_ptr(), nn, gl.data_ptr())
    t0 = lap("labels", t0)
    u, c = torch.unique(gl, return_counts=True)
    t0 = lap("torch.unique (%d components here)" % len(u), t0)
    owner = torch.empty(int(per[0]), dtype=torch.uint8, device=dev)
    cc.owners(gl.data_ptr(), gl.data_ptr(), owner.data_ptr(), 0, owner.data_ptr())
    sk = torch.empty(int(per[0]), dtype=torch.int64, device=dev); sc = torch.empty(int(per[0]), dtype=torch.int32, device=dev)
    send = cc.shard(owner.data_ptr(), sk.data_ptr(), sc.data_ptr())
    t0 = lap("owners + shard", t0)
    cc.close(); cc1.close(); sh.close(); sh1.close()
