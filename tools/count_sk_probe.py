"""Counting stage alone on the BASELINE configs[2] batch (or [reads] [genes] [K]): the partition pipeline (SHN_COUNT_SK=0) against the
super-k-mer path, per-kernel timers of both, tables compared key by key.  PROBE_BITS="20 21 22": bucket grids of the super-k-mer path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from shannon_amd import device
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
K1 = (int(sys.argv[3]) if len(sys.argv) > 3 else 25) + 1
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, torch.device("cuda", 0), exon_len=(80, 600) if K1 == 26 else (80, 5000))
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)


def run(label, reps=2):
    out = None
    for rep in range(reps):
        ctx.timer_reset(); ctx.sync()
        t0 = time.time()
        t = device.count_k1mers(ctx, [d1, d2], K1, True)
        ctx.sync()
        dt = time.time() - t0
        tm = ctx.timers()
        print("%-28s run %d: %.3f s  n=%d  " % (label, rep, dt, len(t)) + "  ".join("%s %.1f" % (k.replace("count.", ""), v[0]) for k, v in sorted(tm.items()) if k.startswith("count.") or k == "table.build"), flush=True)
        if out is None:
            out = t
        else:
            t.close()
    return out


os.environ["SHN_COUNT_SK"] = "0"
t_old = run("partition pipeline")
k0, c0 = t_old.download(); t_old.close()
o = np.argsort(k0); k0, c0 = k0[o], c0[o]
os.environ["SHN_COUNT_SK"] = "1"
for bits in os.environ.get("PROBE_BITS", "0").split():
    if int(bits):
        os.environ["SHN_COUNT_SK_BITS"] = bits
    else:
        os.environ.pop("SHN_COUNT_SK_BITS", None)
    t_new = run("super-k-mers bits=%s" % bits, 3)
    k1, c1 = t_new.download(); t_new.close()
    o = np.argsort(k1)
    print("  equal tables:", bool(np.array_equal(k0, k1[o]) and np.array_equal(c0, c1[o])), flush=True)
