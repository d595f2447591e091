"""Extension stage alone on a benchmark-like input: timers per kernel, and -- from the final walk lengths -- how well the thread
walker's wavefronts are filled (one thread per walk, 64 consecutive ranks per wavefront: a wavefront runs as long as its longest
walk).  usage: ext_probe.py [genes] [reads] [K]   (SHN_DEBUG=1 for the per-round log)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from shannon_amd import device, extension_correction as ec
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 25_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 25
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, dev)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], K + 1, True)
print("k1-mers", len(t), flush=True)
e = ec.Extension(ctx, t, 3); e.close()          # warm
ctx.timer_reset()
t0 = time.time()
e = ec.Extension(ctx, t, 3)
ctx.sync()
print("extension %.3f s, %d walks, %d rounds, %d steps (%d by wavefronts)" % (time.time() - t0, e.n_walks, e.iterations, e.total_steps, e.wave_steps))
for k, v in sorted(ctx.timers().items(), key=lambda kv: -kv[1][0]):
    print("  %-22s %9.2f ms %7d launches" % (k, v[0], v[1]))
nr, nl, tw = e.stats()
ln = np.where(nr == 0xFFFFFFFF, 0, nr.astype(np.int64) + nl.astype(np.int64))
ns = len(ln)
print("walks %d, void %d, steps of the final paths %d, mean %.2f, max %d" % (ns, int((nr == 0xFFFFFFFF).sum()), int(ln.sum()), ln.mean(), ln.max()))
for name, sel in (("first block (ns/8)", ln[:ns // 8]), ("all walks", ln)):
    m = len(sel) // 64 * 64
    w = sel[:m].reshape(-1, 64)
    mx = w.max(axis=1)
    print("  %-20s lanes busy %.4f of the wavefront-steps (sum %d / 64 x sum of maxima %d); wavefronts with max >= 64 steps: %.4f, their share of the wavefront-steps %.3f"
          % (name, w.sum() / max(1, 64 * mx.sum()), int(w.sum()), int(mx.sum()), float((mx >= 64).mean()), float(mx[mx >= 64].sum() / max(1, mx.sum()))))
    q = np.percentile(sel, [50, 90, 99, 99.9, 99.99])
    print("  %-20s walk length percentiles 50/90/99/99.9/99.99: %s" % (name, q.tolist()))
e.close(); t.close(); ctx.close()
