"""Row a28, CPU: the oracle's restatement of cvxopt's interior-point limit (oracle/lp.py: face_center -- unsupported flows from
the exact vertex, supported flows at the analytic centre of the optimal face) against references that are independent of it
and are NOT cvxopt: closed forms, and the log-barrier central path of the same LP followed to mu -> 0 in dense numpy
(tests/lp_barrier_reference.py)."""
import numpy as np
import pytest

from oracle import lp as olp
from lp_barrier_reference import central_path_limit


def _costs(rng, P):
    """generic positive integer cost numerators on the unsupported cells (the oracle's solver wants integers)"""
    m, n = P.shape
    c = rng.integers(1 << 20, 1 << 34, size=(m, n))
    return [[int(c[i, j]) if not P[i, j] else 0 for j in range(n)] for i in range(m)]


def test_two_by_two_closed_form():
    """all four cells supported: x11 = t maximises log t + log(a1 - t) + log(b1 - t) + log(a2 - b1 + t); the root of the derivative by
    bisection.  a = b = (10, 10) -> 5 / 5 / 5 / 5 (the judge's example: four clones, a vertex gives two)."""
    for a1, a2, b1 in ((10.0, 10.0, 10.0), (10.0, 10.0, 15.0), (3.0, 40.0, 7.5), (100.0, 1.0, 50.5), (17.0, 0.25, 0.5)):
        b2 = a1 + a2 - b1
        lo, hi = max(0.0, b1 - a2), min(a1, b1)

        def d(t):
            return 1 / t - 1 / (a1 - t) - 1 / (b1 - t) + 1 / (a2 - b1 + t)
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if d(mid) > 0:
                lo = mid
            else:
                hi = mid
        t = 0.5 * (lo + hi)
        want = [[t, a1 - t], [b1 - t, a2 - b1 + t]]
        st = {}
        got = olp.transport_center([a1, a2], [b1, b2], [[0, 0], [0, 0]], [[1, 1], [1, 1]], st)
        assert np.allclose(got, want, rtol=1e-10, atol=1e-10), (got, want)
        assert st["components"] == 1
    ans, _ = olp.path_decompose([10, 10], [10, 10], [[1, 1], [1, 1]])
    assert ans == [[5.0, 5.0], [5.0, 5.0]]
    ans, _ = olp.path_decompose([10, 10], [10, 10], [[1, 1], [1, 1]], rule="vertex")
    assert sum(1 for r in ans for v in r if v > 0) == 2


def test_complete_support_equal_marginals_is_uniform():
    for m, n in ((2, 3), (3, 3), (4, 6), (7, 5)):
        a = [float(n)] * m
        b = [float(m)] * n
        got = olp.transport_center(a, b, [[0] * n for _ in range(m)], [[1] * n for _ in range(m)])
        assert np.allclose(got, 1.0, rtol=1e-11, atol=0)


@pytest.mark.parametrize("seed", range(40))
def test_equals_the_limit_of_the_log_barrier_central_path(seed):
    """random marginals and supports, 2..6 rows and columns: vertex + classes + Newton on the face (oracle) == the barrier path
    of the full LP followed to mu = 1e-11 (numpy), to 1e-6 of the largest marginal"""
    rng = np.random.default_rng(1000 + seed)
    m, n = int(rng.integers(2, 7)), int(rng.integers(2, 7))
    a = rng.uniform(0.5, 100.0, m)
    b = rng.uniform(0.5, 100.0, n)
    b *= a.sum() / b.sum()
    P = rng.random((m, n)) < [0.3, 0.6, 0.9, 1.0][seed % 4]
    c = _costs(rng, P)
    scale = 1.0 / float(1 << 32)
    st = {}
    got = np.array(olp.transport_center(list(a), list(b), c, P.tolist(), st))
    ref = central_path_limit(a, b, np.array(c, float) * scale)
    assert np.abs(got - ref).max() <= 1e-6 * max(a.max(), b.max()), (m, n, st, np.abs(got - ref).max())


@pytest.mark.parametrize("seed,m,n,dens", [(0, 36, 34, 0.8), (1, 60, 9, 0.6), (2, 7, 66, 0.5)])
def test_nodes_with_more_than_64_rows_and_columns_against_the_barrier_path(seed, m, n, dens):
    """the centre rule beyond 64 nodes (round 5: until then such a node kept its vertex): the same comparison at 70+ nodes"""
    rng = np.random.default_rng(5000 + seed)
    a = rng.uniform(0.5, 100.0, m)
    b = rng.uniform(0.5, 100.0, n)
    b *= a.sum() / b.sum()
    P = rng.random((m, n)) < dens
    c = _costs(rng, P)
    st = {}
    got = np.array(olp.transport_center(list(a), list(b), c, P.tolist(), st))
    assert st.get("components", 0) >= 1 and not st.get("too_large") and not st.get("not_converged")
    ref = central_path_limit(a, b, np.array(c, float) / float(1 << 32))
    assert np.abs(got - ref).max() <= 1e-6 * max(a.max(), b.max()), (m, n, st, np.abs(got - ref).max())


@pytest.mark.parametrize("seed", range(30))
def test_face_properties(seed):
    """marginals kept, unsupported cells keep the vertex's flows, supported cells outside every class stay zero, the KKT condition
    1 / x_ij = u_i + v_j holds on every class that was centred; zero marginals, integer ties and sparse supports included"""
    rng = np.random.default_rng(7000 + seed)
    m, n = int(rng.integers(2, 10)), int(rng.integers(2, 10))
    a = rng.integers(0, 30, m).astype(float)
    if a.sum() == 0:
        a[0] = 3.0
    cuts = np.sort(rng.integers(0, int(a.sum()) + 1, n - 1))
    b = np.diff(np.concatenate([[0], cuts, [a.sum()]])).astype(float)
    P = rng.random((m, n)) < [0.2, 0.5, 0.8][seed % 3]
    c = _costs(rng, P)
    v = olp.transport_vertex(list(a), list(b), c)
    st = {}
    x = np.array(olp.face_center(v, P.tolist(), st))
    v = np.array(v)
    assert np.allclose(x.sum(axis=1), a, atol=1e-9) and np.allclose(x.sum(axis=0), b, atol=1e-9)
    assert np.array_equal(x[~P], v[~P]) and (x >= 0).all()
    assert st.get("not_converged", 0) == 0
    # a cell that moved lies in a class: there the reciprocal flows are a sum of a row and a column potential (rank test on 2x2 minors)
    moved = (x != v)
    for i in range(m):
        for i2 in range(i + 1, m):
            for j in range(n):
                for j2 in range(j + 1, n):
                    if moved[i, j] and moved[i, j2] and moved[i2, j] and moved[i2, j2]:
                        k = 1 / x[i, j] - 1 / x[i, j2] - 1 / x[i2, j] + 1 / x[i2, j2]
                        assert abs(k) <= 1e-8 * max(1 / x[i, j], 1 / x[i, j2], 1 / x[i2, j], 1 / x[i2, j2])


def test_vertex_rule_is_kept_behind_the_switch():
    rng = np.random.default_rng(3)
    for t in range(20):
        m, n = int(rng.integers(2, 6)), int(rng.integers(2, 6))
        a = [float(v) for v in rng.integers(1, 30, m)]
        b = [float(v) for v in rng.integers(1, 30, n)]
        P = (rng.random((m, n)) < 0.5).astype(int).tolist()
        old = olp.path_decompose(a, b, P, seed=9, pid=t, rule="vertex")
        via_solver = olp.path_decompose(a, b, P, seed=9, pid=t, solver=olp.transport_vertex)
        assert old == via_solver
