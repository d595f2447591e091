"""Repeat the played-ranks labelling on one table and compare every run's global labels with the first (round 5: the component tables
of a 2-rank strand-specific job differed between runs while the owned shards were identical).
usage: python tools/stress_cc_r05.py <repeats> <W> <ss 0/1> [tag]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from shannon_amd import device, synth
import test_cc_shards_gpu as T
from cc_reference import reference_labels, same_partition
rep, W, ss = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1"
tag = sys.argv[4] if len(sys.argv) > 4 else ""
(q1, q2), _ = synth.make_dataset(12000, 12, seed=4)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, q1), device.Reads.from_codes(ctx, q2)
table = device.count_k1mers_strand_specific(ctx, d1, d2, 26) if ss else device.count_k1mers(ctx, [d1, d2], 26, True)
first, bad, t0 = None, 0, time.time()
ref = None
for i in range(rep):
    out, got = T.play_ranks(ctx, table, W, 26, table.canonical)
    keys = np.concatenate([o[0] for o in out]); gl = np.concatenate([o[1] for o in out]); ow = np.concatenate([o[2] for o in out])
    if first is None:
        first = (keys, gl, ow)
        ref = reference_labels(keys, 26, table.canonical)
        print(tag, "first run equals the reference partition:", same_partition(gl, ref), "components", len(np.unique(ref)))
    else:
        if not (np.array_equal(keys, first[0]) and np.array_equal(gl, first[1])):
            bad += 1
            d = np.nonzero(gl != first[1])[0]
            print(tag, "run", i, "labels differ at", len(d), "k1-mers; partition still the reference's:", same_partition(gl, ref))
print(tag, "%d runs, %d differ, %.0f s" % (rep, bad, time.time() - t0))
