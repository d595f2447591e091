"""NOT cvxopt.  An independent reference for the limit an interior-point method converges to on the trial LPs of
path_decompose_sparse.py:100-117: the log-barrier central path  x(mu) = argmin c.x / mu - sum log x  s.t. A x = b  followed to
mu -> 0 with dense numpy linear algebra.  Its limit is, by definition, the analytic centre of the optimal face -- what
oracle/lp.py:face_center computes combinatorially (vertex + classes of the residual digraph + Newton on the face).  Test
infrastructure only; it shares no code with the oracle."""
import numpy as np


def _constraints(m, n):
    """A of path_decompose_sparse.py:74-84: rows 0..m-1 the row sums, rows m..m+n-2 the sums of the first n-1 columns;
    variable k = j*m + i."""
    A = np.zeros((m + n - 1, m * n))
    for i in range(m):
        for j in range(n):
            A[i, j * m + i] = 1.0
    for j in range(n - 1):
        for i in range(m):
            A[m + j, j * m + i] = 1.0
    return A


def _interior_start(A, b):
    """strictly positive x with A x = b: analytic centre of the whole polytope by infeasible-start Newton"""
    N = A.shape[1]
    x = np.full(N, max(b.max(), 1e-300) / max(1, N // max(1, A.shape[0])))
    nu = np.zeros(A.shape[0])
    for _ in range(200):
        rd = -1.0 / x + A.T @ nu
        rp = A @ x - b
        r = np.sqrt((rd ** 2).sum() + (rp ** 2).sum())
        if r < 1e-11:
            break
        X2 = x * x
        M = (A * X2) @ A.T
        w = np.linalg.solve(M, 2 * (A @ x) - b)
        dx = x - X2 * (A.T @ w)
        dnu = w - nu
        t = 1.0
        while (x + t * dx <= 0).any():
            t *= 0.5
        while True:
            xn, nun = x + t * dx, nu + t * dnu
            rn = np.sqrt(((-1.0 / xn + A.T @ nun) ** 2).sum() + ((A @ xn - b) ** 2).sum())
            if rn <= (1 - 0.01 * t) * r or t < 1e-14:
                break
            t *= 0.5
        x, nu = xn, nun
    return x


def central_path_limit(a, b, c, mu_end=1e-11):
    """a (m), b (n) balanced marginals, all > 0; c (m x n) costs >= 0.  Returns the m x n limit of the central path."""
    a, b, c = np.asarray(a, float), np.asarray(b, float), np.asarray(c, float)
    m, n = len(a), len(b)
    A = _constraints(m, n)
    rhs = np.concatenate([a, b[:n - 1]])
    cv = np.array([c[k % m, k // m] for k in range(m * n)])
    x = _interior_start(A, rhs)
    mu = max(1.0, float(cv.max())) * float(x.max())
    while mu > mu_end:
        mu *= 0.2
        for _ in range(100):
            g = cv / mu - 1.0 / x
            X2 = x * x
            M = (A * X2) @ A.T
            w = np.linalg.solve(M, -(A * X2) @ g)
            dx = -X2 * (g + A.T @ w)
            lam2 = float(-(g @ dx))                     # Newton decrement squared
            if not lam2 > 1e-24:
                break
            # damped Newton for a self-concordant function: no function values (c.x / mu ~ 1e13 would drown them in rounding),
            # the step stays inside the Dikin ellipsoid, so x stays positive
            lam = np.sqrt(lam2)
            t = 1.0 if lam <= 0.25 else 1.0 / (1.0 + lam)
            while (x + t * dx <= 0).any() and t > 1e-30:
                t *= 0.5
            x = x + t * dx
    return np.array([[x[j * m + i] for j in range(n)] for i in range(m)])
