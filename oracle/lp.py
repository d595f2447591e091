"""Sparse-flow node decomposition oracle (row a28).  Test infrastructure (see
oracle/__init__.py).  Restates path_decompose_sparse.py:15-193.

THIRD-PARTY ARITHMETIC, PARITY UNPINNED: the reference solves each trial LP with
cvxopt.solvers.lp (interior point; cvxopt is imported at path_decompose_sparse.py:29 and
algorithm_SF.py:12, no version pinned anywhere, not vendored, not installable here) and draws
unseeded numpy.random.normal costs (:107).  On a degenerate optimal face cvxopt returns the
analytic centre; this build *defines* its own rule instead:

  * costs: counter-based generator, |Irwin-Hall(12) - 6| from splitmix64 hashes of
    (seed, problem id, trial, cell) -- integer arithmetic only, so CPU and GPU agree bit for bit;
  * LP: the transportation problem  min c.x, row sums a, column sums b, x >= 0  is solved to an
    exact *vertex* by successive shortest paths with Jacobi Bellman-Ford rounds and
    lowest-index tie-breaks (spec below; the HIP kernel implements the identical sequence).

Everything around the LP (balancing, scaling, thresholds, trial selection, top-`sparsity`
truncation) follows the reference line by line and is pinned against it in tests/golden.
"""
import math
import numpy as np

M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15
INF = 1 << 62


def splitmix64(x):
    x = (x + GOLD) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def cell_cost(seed, pid, trial, cell):
    """|N(0,1)| stand-in, as an INTEGER numerator: |sum of 12 uniform 32-bit draws - 6*2^32|
    (cost = value / 2^32).  Distances in the solver are sums/differences of these integers, so
    they are exact (no rounding => no spurious negative cycles from (d+c)-c != d)."""
    seed, pid, trial, cell = int(seed), int(pid), int(trial), int(cell)
    h = splitmix64((seed ^ (((pid + 1) * 0xD1B54A32D192ED03) & M64)) & M64)
    h = splitmix64(h ^ (((trial + 1) * 0xAEF17502108EF2D9) & M64))
    h = splitmix64(h ^ (((cell + 1) * 0x8CB92BA72F3D8DD7) & M64))
    s = 0
    st = h
    for _ in range(6):
        st = splitmix64(st)
        s += (st & 0xFFFFFFFF) + (st >> 32)
    return abs(s - 6 * (1 << 32))


def trial_costs(seed, pid, trial, mn):
    return [cell_cost(seed, pid, trial, k) for k in range(mn)]


def transport_vertex(a, b, c):
    """Exact vertex of  min sum c[i][j] x[i][j]  s.t. row sums a, col sums b, x>=0.
    a: m floats, b: n floats (sum(a) ~= sum(b)), c: m x n list of lists of non-negative INTEGERS
    (cost numerators, see cell_cost).  Returns x (m x n) floats.

    Spec (shared with shannon_amd/csrc/lp_kernels.hip):
      ra=a, rb=b, x=0.  Repeat:
        S={i: ra[i]>0}, T={j: rb[j]>0}; stop if either is empty.
        ds[i]=0 (i in S) else INF, ps[i]=-1; dt[j]=INF, pt[j]=-1.
        Up to m+n rounds: (F) for every j: best=min_i ds[i]+c[i][j] (lowest i on ties, only ds[i]<INF);
        if best<dt[j]: dt[j]=best, pt[j]=i.  (B) for every i: best=min over j with x[i][j]>0 and
        dt[j]<INF of dt[j]-c[i][j] (lowest j on ties); if best<ds[i]: ds[i]=best, ps[i]=j.
        Stop rounds when nothing changed.
        t = argmin_{j in T} dt[j] (lowest j on ties); stop if dt[t]>=INF.
        Trace t<-pt<-ps<-... back to the first source with ps==-1 (=s); delta=min(ra[s], rb[t],
        x on the backward arcs of the path); forward arcs += delta, backward arcs -= delta,
        ra[s]-=delta, rb[t]-=delta.
    """
    m, n = len(a), len(b)
    x = [[0.0] * n for _ in range(m)]
    ra, rb = [float(v) for v in a], [float(v) for v in b]
    for _it in range(4 * (m + n) + m * n + 16):
        S = [i for i in range(m) if ra[i] > 0]
        T = [j for j in range(n) if rb[j] > 0]
        if not S or not T:
            break
        ds = [INF] * m
        ps = [-1] * m
        for i in S:
            ds[i] = 0
        dt = [INF] * n
        pt = [-1] * n
        for _r in range(m + n):
            changed = False
            for j in range(n):
                best, bi = INF, -1
                for i in range(m):
                    if ds[i] < INF:
                        v = ds[i] + c[i][j]
                        if v < best:
                            best, bi = v, i
                if bi >= 0 and best < dt[j]:
                    dt[j], pt[j] = best, bi
                    changed = True
            for i in range(m):
                best, bj = INF, -1
                for j in range(n):
                    if x[i][j] > 0 and dt[j] < INF:
                        v = dt[j] - c[i][j]
                        if v < best:
                            best, bj = v, j
                if bj >= 0 and best < ds[i]:
                    ds[i], ps[i] = best, bj
                    changed = True
            if not changed:
                break
        t, bd = -1, INF
        for j in T:
            if dt[j] < bd:
                bd, t = dt[j], j
        if t < 0:
            break
        # trace back
        fwd, bwd = [], []
        j = t
        delta = rb[t]
        guard = 0
        while True:
            i = pt[j]
            fwd.append((i, j))
            if ps[i] < 0:
                s = i
                break
            j = ps[i]
            bwd.append((i, j))
            if x[i][j] < delta:
                delta = x[i][j]
            guard += 1
            if guard > m + n + 2:
                raise RuntimeError("predecessor loop")
        if ra[s] < delta:
            delta = ra[s]
        for i, j in fwd:
            x[i][j] += delta
        for i, j in bwd:
            x[i][j] -= delta
        ra[s] -= delta
        rb[t] -= delta
    return x


def n_trials(m, n):
    """path_decompose_sparse.py:100."""
    return int(round(min(2 * m * n * max(m, n), 100)))


def path_decompose(a, b, P, seed=0, pid=0, sparsity=10, solver=None, cost_fn=None):
    """path_decompose_sparse.py:15-193 with overwrite_norm=False, use_GLPK=False.
    a, b: lists of floats; P: m x n 0/1 (list of lists); returns (answer m x n list of lists, non_unique).
    """
    m, n = len(a), len(b)
    if m == 0 or n == 0:
        return [], 0                                            # :38-39
    if m == 1:
        return [[float(v) for v in b]], 0                       # :41-43
    if n == 1:
        return [[float(v)] for v in a], 0                       # :44-46
    sa = 0.0
    for v in a:
        sa += v
    sb = 0.0
    for v in b:
        sb += v
    if sa <= 0 or sb <= 0:
        return [[0.0] * n for _ in range(m)], 0                 # :48-50
    a = [float(v) for v in a]
    b = [float(v) for v in b]
    if sa > sb:                                                 # :64-69
        const = sa - sb
        b = [k + const * k / sb for k in b]
    else:
        const = sb - sa
        a = [k + const * k / sa for k in a]
    p = [0.0] * (m * n)                                         # p[j*m+i] = 1 - P[i][j]
    for i in range(m):
        for j in range(n):
            p[j * m + i] = 1.0 - float(P[i][j])
    z = a + b
    rhs = z[:m + n - 1]
    weight = 0.0
    for v in a:
        weight += abs(v)
    tol = 0.001 * weight                                        # :92-94
    fac = 0.4                                                   # :95-96 (both factors)
    scale = max(max(rhs), 1e-100) * 0.01                        # :97
    rs = [v / scale for v in rhs]
    a_s = rs[:m]
    b_s = rs[m:]
    tot = 0.0
    for v in a_s:
        tot += v
    for v in b_s:
        tot -= v
    b_s = b_s + [tot if tot > 0 else 0.0]                       # implied last column (:82)
    trials = n_trials(m, n)
    curr_min = m * n + 1
    curr_ans = None
    curr_mult = 0
    curr_on_unknown = 0.0
    mn = m * n
    for ctr in range(trials):
        cc = (cost_fn or trial_costs)(seed, pid, ctr, mn)
        c = [[(cc[j * m + i] if p[j * m + i] > 0 else 0) for j in range(n)] for i in range(m)]
        xs = (solver or transport_vertex)(a_s, b_s, c)
        temp = [xs[k % m][k // m] * scale for k in range(mn)]   # temp_sol[j*m+i]
        for i in range(m):
            for j in range(n):
                k = j * m + i
                thr = fac * min(a[i], b[j])
                if temp[k] < thr or temp[k] < tol or temp[k] < 0:
                    temp[k] = 0.0
        s = 0
        for k in range(mn):
            if p[k] > 0 and temp[k] != 0:
                s += 1
        dot = 0.0
        for k in range(mn):
            dot += p[k] * temp[k]
        if s < curr_min:                                        # :147-151
            curr_min, curr_ans, curr_mult, curr_on_unknown = s, temp, 0, dot
        elif s == curr_min:                                     # :153-161
            d2 = 0.0
            for k in range(mn):
                d2 += (curr_ans[k] - temp[k]) ** 2
            if math.sqrt(d2) > tol:
                curr_mult += 1
            st = 0.0
            for v in temp:
                st += v
            sc = 0.0
            for v in curr_ans:
                sc += v
            if (abs(st - sc) < tol and dot < curr_on_unknown) or st > sc:
                curr_ans, curr_on_unknown = temp, dot
    answer = [[float(curr_ans[j * m + i]) for j in range(n)] for i in range(m)]
    non_unique = 1 if curr_mult > 1 else 0
    if sparsity and m * n > sparsity:                           # :180-192
        cells = [((i, j), answer[i][j]) for i in range(m) for j in range(n)]
        cells = sorted(cells, key=lambda t: t[1])[::-1][:sparsity]
        new = [[0.0] * n for _ in range(m)]
        for (i, j), v in cells:
            new[i][j] = v
        answer = new
    return answer, non_unique
