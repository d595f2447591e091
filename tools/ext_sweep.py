"""Extension stage under a list of environment settings on one table (BASELINE configs[2] by default): total, rounds and the main
kernels.  usage: ext_sweep.py "A=1 B=2" "C=3" ...   (every argument one setting; "" = defaults)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from shannon_amd import device, extension_correction as ec
genes, reads, K = int(os.environ.get("SWEEP_GENES", 20000)), int(os.environ.get("SWEEP_READS", 100_000_000)), 25
dev = torch.device("cuda", 0)
r1, r2 = bench.gen_reads(reads // 2, 20240501, genes, dev)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
t = device.count_k1mers(ctx, [d1, d2], K + 1, True)
e = ec.Extension(ctx, t, 3); e.close()          # warm
for setting in sys.argv[1:] or [""]:
    kv = dict(x.split("=") for x in setting.split())
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    best = None
    for rep in range(2):
        ctx.timer_reset()
        t0 = time.time()
        e = ec.Extension(ctx, t, 3)
        ctx.sync()
        dt = time.time() - t0
        T = ctx.timers()
        row = (dt, e.iterations, e.total_steps, {k: round(T[k][0], 1) for k in ("extend.walk_thread", "extend.walk_wave", "extend.mark", "extend.prepare") if k in T})
        e.close()
        if best is None or dt < best[0]:
            best = row
    print("%-60s %.3f s  rounds %d  steps %d  %s" % (setting or "(defaults)", best[0], best[1], best[2], best[3]), flush=True)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
