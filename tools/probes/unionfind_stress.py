"""the lock-free union-find alone (shn_cc_solve: every edge united once): long chains and random forests, shuffled, against scipy"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from shannon_amd import device
ctx = device.Context(0)
rng = np.random.default_rng(1)
bad = 0
for rep in range(40):
    n = 2_000_000
    kind = rep % 3
    if kind == 0:      # chains of 1000
        a = np.arange(n - 1); keep = (a % 1000) != 999; e = np.stack([a[keep], a[keep] + 1], 1)
    elif kind == 1:    # random sparse graph
        e = rng.integers(0, n, (n // 2, 2))
    else:              # a few huge chains + random links
        a = np.arange(n - 1); keep = (a % 500000) != 499999; e = np.concatenate([np.stack([a[keep], a[keep] + 1], 1), rng.integers(0, n, (1000, 2))])
    e = e[rng.permutation(len(e))]
    flip = rng.random(len(e)) < 0.5
    e[flip] = e[flip][:, ::-1]
    ge = torch.as_tensor(e.astype(np.int64).reshape(-1), device="cuda")
    E = len(e)
    ids = torch.empty(2 * E, dtype=torch.int64, device="cuda"); labs = torch.empty(2 * E, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nn = device.ComponentShards.solve(ctx, ge.data_ptr(), E, n + 1, ids.data_ptr(), labs.data_ptr())
    ids_h, labs_h = ids[:nn].cpu().numpy(), labs[:nn].cpu().numpy()
    _, ref = connected_components(coo_matrix((np.ones(E, np.int8), (e[:, 0], e[:, 1])), shape=(n, n)), directed=False)
    first = np.full(ref.max() + 1, n, dtype=np.int64); np.minimum.at(first, ref, np.arange(n))
    want = first[ref][ids_h]
    ok = np.array_equal(labs_h, want)
    if not ok:
        bad += 1
        print("rep", rep, "kind", kind, "WRONG at", int((labs_h != want).sum()), "of", nn)
print("40 graphs,", bad, "wrong")
