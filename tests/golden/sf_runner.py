"""Runs the translated algorithm_SF.py script for one component with the stub cvxopt, the
oracle's restatement of the interior-point limit (oracle/lp.py: transport_center) plugged into
solvers.lp, and numpy.random.normal replaced by the oracle's counter-based cost generator.  GOLDEN-VECTOR HARNESS ONLY.
usage: sf_runner.py <tref> <seed> <comp_id_for_rng> <comp> <prefix>"""
import sys, os, runpy
import numpy as np
tref, seed, comp_rng = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sys.argv = ["algorithm_SF.py"] + sys.argv[4:]
import cvxopt
from oracle import lp as olp
import path_decompose_sparse as pds

state = {"call": -1, "trial": 0}
_orig = pds.path_decompose


def wrapped(*a, **k):
    state["call"] += 1
    state["trial"] = 0
    return _orig(*a, **k)


pds.path_decompose = wrapped


def fake_normal(mu, sigma, size):
    mn = size[0]
    pid = (comp_rng << 20) + state["call"]
    v = np.array(olp.trial_costs(seed, pid, state["trial"], mn), dtype=float) / float(1 << 32)
    state["trial"] += 1
    return v.reshape(size)


np.random.normal = fake_normal


def lp_impl(c, A, b):
    # recover m, n from A ((m+n-1) x mn, x[j*m+i]); rows 0..m-1 are the row-sum constraints
    mn = A.shape[1]
    n = int(round(A[0].sum()))
    m = mn // n
    a_s = [float(v) for v in b.reshape(-1)[:m]]
    b_s = [float(v) for v in b.reshape(-1)[m:]]
    tot = 0.0
    for v in a_s:
        tot += v
    for v in b_s:
        tot -= v
    b_s.append(tot if tot > 0 else 0.0)
    cf = c.reshape(-1)
    ci = [[int(round(cf[j * m + i] * (1 << 32))) for j in range(n)] for i in range(m)]
    if os.environ.get("SHN_LP_RULE") == "vertex":
        x = olp.transport_vertex(a_s, b_s, ci)
    else:                                        # the interior-point limit: supported (zero-cost) cells at the centre of the optimal face
        x = olp.transport_center(a_s, b_s, ci, [[cf[j * m + i] == 0 for j in range(n)] for i in range(m)])
    return np.array([x[k % m][k // m] for k in range(mn)], dtype=float).reshape(-1, 1)


cvxopt.solvers.lp_impl = lp_impl
runpy.run_path(os.path.join(tref, "algorithm_SF.py"), run_name="__main__")
