"""Counting stage alone on the BASELINE configs[2] batch (or --reads / --genes), once per setting of the level split:
SHN_COUNT_B1 (bits of level 1) and SHN_COUNT_LEVELS (2 / 3).  usage: count_probe.py [reads] [genes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from shannon_amd import device
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, dev)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], 26, True); n0 = len(t); t.close()      # warm (buffers, the one-pass path's verdict)
SETTINGS = [tuple(int(x) for x in a.split(",")) for a in os.environ.get("PROBE_SETTINGS", "0,2,0 11,2,0 10,2,0 9,2,0 8,2,0 8,3,0 10,3,0").split()]
for b1, lv, bits in SETTINGS:                      # bits: SHN_COUNT_BITS (buckets of the chunk tables), 0 = the estimate's
    if b1: os.environ["SHN_COUNT_B1"] = str(b1)
    else: os.environ.pop("SHN_COUNT_B1", None)
    if bits: os.environ["SHN_COUNT_BITS"] = str(bits)
    else: os.environ.pop("SHN_COUNT_BITS", None)
    os.environ["SHN_COUNT_LEVELS"] = str(lv)
    ctx.timer_reset(); ctx.sync()
    t0 = time.time()
    t = device.count_k1mers(ctx, [d1, d2], 26, True)
    ctx.sync()
    dt = time.time() - t0
    assert len(t) == n0
    t.close()
    tm = ctx.timers()
    print("b1=%2d levels=%d bits=%2d: %.3f s  " % (b1, lv, bits, dt) + "  ".join("%s %.1f" % (k.replace("count.", ""), v[0]) for k, v in sorted(tm.items()) if k.startswith("count.") or k == "table.build"), flush=True)
