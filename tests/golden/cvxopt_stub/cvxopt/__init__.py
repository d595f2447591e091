"""Minimal stand-in for the third-party `cvxopt` module -- GOLDEN-VECTOR HARNESS ONLY.

cvxopt is not installed here and cannot be fetched.  This stub only lets the (translated)
reference files import and run: `matrix` is a numpy-backed column-major dense matrix with
the handful of operations algorithm_SF.py / path_decompose_sparse.py use, and `solvers.lp`
delegates to a pluggable function (set by the harness to the oracle's pinned transportation
solver).  It is NOT cvxopt and does not reproduce its interior-point iterates.
"""
import numpy as _np


class matrix(object):
    def __init__(self, x=0.0, size=None):
        if isinstance(x, matrix):
            arr = x.a.copy()
        elif isinstance(x, _np.ndarray):
            arr = _np.array(x, dtype=float)
            if arr.ndim == 1:
                arr = arr.reshape(-1, 1)
        elif isinstance(x, (list, tuple)):
            arr = _np.array(x, dtype=float).reshape(-1, 1)
        else:
            arr = _np.array([[float(x)]])
        if size is not None:
            if arr.size == 1:
                arr = _np.full(size, float(arr.flat[0]))
            else:
                arr = arr.reshape(-1, order="F").reshape(size, order="F")
        self.a = arr

    @property
    def size(self):
        return self.a.shape

    def __len__(self):
        return self.a.size

    def __iter__(self):
        return iter(self.a.reshape(-1, order="F").tolist())

    def _lin(self, k):
        m = self.a.shape[0]
        return (k % m, k // m)

    def __getitem__(self, k):
        if isinstance(k, tuple):
            r = self.a[k]
            if isinstance(r, _np.ndarray):
                return matrix(r.reshape(-1, 1) if r.ndim == 1 else r)
            return float(r)
        if isinstance(k, slice):
            return matrix(self.a.reshape(-1, order="F")[k])
        return float(self.a[self._lin(int(k))])

    def __setitem__(self, k, v):
        if isinstance(k, tuple):
            self.a[k] = v
        else:
            self.a[self._lin(int(k))] = v

    def __truediv__(self, s):
        return matrix(self.a / s)

    def __mul__(self, s):
        return matrix(self.a * s)

    __rmul__ = __mul__

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)


def spmatrix(v, I, J, size=None):
    I, J = list(I), list(J)
    n = (max(I) + 1, max(J) + 1) if size is None else size
    m = matrix(0.0, n)
    for i, j in zip(I, J):
        m.a[i, j] = v
    return m


def spdiag(*a, **k):
    raise NotImplementedError("cvxopt stub: spdiag is only used by dead code (use_smoothing=False)")


sparse = spdiag


class _Solvers(object):
    options = {}
    lp_impl = None

    def lp(self, c, G, h, A=None, b=None, solver=None):
        if self.lp_impl is None:
            raise RuntimeError("cvxopt stub: no LP implementation plugged in")
        return {"x": matrix(self.lp_impl(_np.array(c), _np.array(A), _np.array(b))), "status": "optimal"}

    def coneqp(self, *a, **k):
        raise NotImplementedError("cvxopt stub: coneqp is dead code in the hot path")


solvers = _Solvers()
