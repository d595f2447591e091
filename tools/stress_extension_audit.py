import os, sys, time
os.environ["SHN_EXT_AUDIT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shannon_amd import device, synth, extension_correction as ec
ctx = device.Context(0)
n = 0
t0 = time.time()
for seed in range(int(sys.argv[1])):
    ng = [3, 20, 60][seed % 3]
    (r1, r2), _ = synth.make_dataset(60000, ng, seed=100 + seed)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    for rep in range(int(sys.argv[2])):
        for W in (1, 2, 3):
            for rk in range(W):
                e = ec.Extension(ctx, t, 3, 100000, shard=(W, rk)); e.close(); n += 1
    t.close() if hasattr(t, "close") else None
print("extensions", n, "sec", round(time.time() - t0, 1))
