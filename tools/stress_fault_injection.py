import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shannon_amd import device, synth, extension_correction as ec
ctx = device.Context(0)
for seed, ng in ((78, 20), (5, 3), (9, 60)):
    (r1, r2), _ = synth.make_dataset(60000, ng, seed=seed)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    os.environ.pop("SHN_EXT_FAULT", None)
    os.environ["SHN_EXT_AUDIT"] = "2"
    e = ec.Extension(ctx, t, 3, 100000); ref = e.stats(); it0 = e.iterations; e.close()
    for f in (1, 2, 3, 4, 6, 9, 12):
        os.environ["SHN_EXT_FAULT"] = str(f)
        e = ec.Extension(ctx, t, 3, 100000); st = e.stats(); it = e.iterations; e.close()
        same = all(np.array_equal(a, b) for a, b in zip(ref, st))
        print("seed", seed, "fault at round", f, "rounds", it0, "->", it, "same result:", same)
