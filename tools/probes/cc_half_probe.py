"""which unions does the labelling lose when every edge is asked from one end only (SHN_CC_HALF=1)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from shannon_amd import device, synth
from cc_reference import reference_labels, same_partition, which_keys, low_complexity
(q1, q2), _ = synth.make_dataset(12000, 12, seed=4)
ctx = device.Context(0)
d1, d2 = device.Reads.from_codes(ctx, q1), device.Reads.from_codes(ctx, q2)
table = device.count_k1mers(ctx, [d1, d2], 26, True)
keys, _ = table.download()
n = len(keys)


def labels():
    cc = device.ComponentShards(ctx, table, 1, 0)
    gl = torch.empty(n, dtype=torch.int64, device="cuda")
    cc.labels(0, 0, 0, 0, gl.data_ptr())
    cc.close()
    return gl.cpu().numpy()


from shannon_amd import _lib
import ctypes as C
os.environ["SHN_CC_DEBUG"] = "1"


def hits():
    a = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.lib().shn_debug_cc_counters(a.ctypes.data_as(_lib.u64p), 1))
    return int(a[0])


os.environ["SHN_CC_HALF"] = "0"
hits()
good = labels()
print("half=0: look-ups that found their key:", hits())
ref = reference_labels(keys, 26, True)
print("half=0 equals the reference:", same_partition(good, ref), "components", len(np.unique(good)))
os.environ["SHN_CC_HALF"] = "1"
order = np.argsort(keys); sk = keys[order]
alive = ~low_complexity(sk, 26)
for rep in range(30):
    lab = labels()
    print("half=1 rep", rep, "look-ups that found their key:", hits(), "labels equal:", np.array_equal(lab, good))
    if np.array_equal(lab, good):
        continue
    d = np.nonzero(lab != good)[0]
    print("rep", rep, "differs at", len(d), "k1-mers")
    for i in d[:3]:
        x = keys[i]
        print("  k1-mer index", i, "key %013x" % x, "good root", good[i], "got root", lab[i], "root key good %013x got %013x" % (keys[good[i]], keys[lab[i]]))
        # its neighbours in the table
        j = np.searchsorted(sk, x)
        for which, (y, ok) in enumerate(which_keys(sk[j:j + 1], 26, True)):
            if not ok[0]:
                continue
            pos = np.searchsorted(sk, y[0])
            if pos < len(sk) and sk[pos] == y[0] and alive[pos]:
                t = order[pos]
                print("     which", which, "neighbour index", t, "key %013x" % y[0], "larger" if y[0] > x else "smaller", "good root", good[t], "got root", lab[t])
    if rep > 6:
        break
