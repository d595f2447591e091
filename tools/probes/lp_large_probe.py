"""one 40 x 30 decomposition (more than 64 rows + columns) through the LP kernels: census and the first differing trial"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_lp_gpu as T
from shannon_amd import device, sparse_flow
from oracle import lp as olp
ctx = device.Context(0)
reqs = T._requests()[1:2]
q = reqs[0]
ctx.lp_stats(reset=True)
xs = sparse_flow.solve_batch(ctx, reqs, 77)
print(ctx.lp_stats())
x = xs[0]
sup = [[not (q.p[j * q.m + i] > 0) for j in range(q.n)] for i in range(q.m)]
for t in range(3):
    cc = olp.trial_costs(77, q.pid, t, q.m * q.n)
    c = [[(cc[j * q.m + i] if q.p[j * q.m + i] > 0 else 0) for j in range(q.n)] for i in range(q.m)]
    v = olp.transport_vertex(q.a_s, q.b_s, c)
    o = {}
    ref = olp.face_center(v, sup, o)
    fv = np.array([v[k % q.m][k // q.m] for k in range(q.m * q.n)])
    fr = np.array([ref[k % q.m][k // q.m] for k in range(q.m * q.n)])
    print(t, o, "gpu==vertex", np.array_equal(fv, x[:, t]), "gpu==centre", np.array_equal(fr, x[:, t]), "max|gpu-centre|", np.abs(fr - x[:, t]).max(),
          "nonzero gpu/ref", (x[:, t] > 0).sum(), (fr > 0).sum())
