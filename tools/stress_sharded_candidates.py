import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shannon_amd import device, synth, extension_correction as ec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
(r1, r2), _ = synth.make_dataset(60000, 20, seed=78)
ctx = device.Context(0)
t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
def cands(shard):
    out = []
    ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, shard=shard, merge=lambda loc: (out.extend(loc), list(loc))[1])
    return out
whole = cands((1, 0))
key = lambda c: (-c[0], c[1])
bad = 0
for i in range(N):
    merged = sorted((c for r in range(W) for c in cands((W, r))), key=key)
    if merged != whole:
        bad += 1
        a = {(c[0], c[1]): c[2] for c in whole}; b = {(c[0], c[1]): c[2] for c in merged}
        only_a = [k for k in a if k not in b]; only_b = [k for k in b if k not in a]; diff = [k for k in a if k in b and a[k] != b[k]]
        print("run", i, "only whole", only_a[:3], "only sharded", only_b[:3], "different string", [(k, len(a[k]), len(b[k])) for k in diff[:3]])
    w2 = cands((1, 0))
    if w2 != whole:
        a = {(c[0], c[1]): c[2] for c in whole}; b = {(c[0], c[1]): c[2] for c in w2}
        diff = [k for k in a if k in b and a[k] != b[k]]
        for k in diff[:2]:
            x, y = a[k], b[k]
            pos = [j for j in range(min(len(x), len(y))) if x[j] != y[j]]
            print("run", i, "UNSHARDED differs", k, len(x), len(y), "positions", pos[:10], len(pos), "chars", [(x[j], y[j]) for j in pos[:6]], "only", [kk for kk in a if kk not in b][:2], [kk for kk in b if kk not in a][:2])
print("W", W, "bad", bad, "of", N)
