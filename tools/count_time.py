"""Counting stage alone on the BASELINE configs[2] batch, timers per kernel (no comparison: tools/count_sk_probe.py does that).
SHN_HIP_LIB picks the build; one process can only load one, so variants are separate runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from shannon_amd import device
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
r1, r2 = bench.gen_reads(reads // 2, 20240501, 20000, torch.device("cuda", 0))
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
for rep in range(3):
    ctx.timer_reset(); ctx.sync()
    t0 = time.time()
    t = device.count_k1mers(ctx, [d1, d2], 26, True)
    ctx.sync()
    dt = time.time() - t0
    tm = ctx.timers()
    print("%-24s run %d: %.3f s  n=%d  " % (os.path.basename(os.environ.get("SHN_HIP_LIB", "default")), rep, dt, len(t)) +
          "  ".join("%s %.1f" % (k.replace("count.", ""), v[0]) for k, v in sorted(tm.items()) if k.startswith("count.")), flush=True)
    t.close()
