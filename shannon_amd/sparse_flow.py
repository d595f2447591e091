"""Sparse-flow transcript reconstruction -- host mirror of the reference's algorithm_SF.py /
path_decompose_sparse.py (rows a25-a30) over the batched HIP LP kernel (csrc/lp.hip).

algorithm_SF.py is a script whose module body runs on import (algorithm_SF.py:858-934); here it
is a function over in-memory component tables.  Every component is a coroutine that yields its
next node decomposition; the driver gathers the pending decompositions of ALL components and
solves their randomized trials in one device batch (the reference runs one process per
component and <=100 serial cvxopt calls per node, run_MB_SF_fn.py:239-254).
"""
import ctypes as C
import math
import sys
import numpy as np
from . import _lib

PATH_SPARSITY = 10      # algorithm_SF.py:31


def n_trials(m, n):
    """path_decompose_sparse.py:100."""
    return int(round(min(2 * m * n * max(m, n), 100)))


class _Req(object):
    """One path_decompose call that needs LP trials (non-trivial case)."""
    __slots__ = ("a", "b", "p", "m", "n", "scale", "a_s", "b_s", "tol", "pid", "sparsity", "trials")


def prepare(a, b, P, pid, sparsity):
    """path_decompose up to the trial loop (path_decompose_sparse.py:33-100).  Returns either
    ('done', answer, non_unique) for the closed-form cases or ('lp', _Req)."""
    m, n = len(a), len(b)
    if m == 0 or n == 0:
        return "done", [], 0
    if m == 1:
        return "done", [[float(v) for v in b]], 0
    if n == 1:
        return "done", [[float(v)] for v in a], 0
    sa = 0.0
    for v in a:
        sa += v
    sb = 0.0
    for v in b:
        sb += v
    if sa <= 0 or sb <= 0:
        return "done", [[0.0] * n for _ in range(m)], 0
    a = [float(v) for v in a]
    b = [float(v) for v in b]
    if sa > sb:
        const = sa - sb
        b = [k + const * k / sb for k in b]
    else:
        const = sb - sa
        a = [k + const * k / sa for k in a]
    q = _Req()
    q.m, q.n, q.a, q.b, q.pid, q.sparsity = m, n, a, b, pid, sparsity
    q.p = np.array([1.0 - float(P[i][j]) for j in range(n) for i in range(m)])      # p[j*m+i]
    rhs = (a + b)[:m + n - 1]
    weight = 0.0
    for v in a:
        weight += abs(v)
    q.tol = 0.001 * weight
    q.scale = max(max(rhs), 1e-100) * 0.01
    rs = [v / q.scale for v in rhs]
    q.a_s = rs[:m]
    bs = rs[m:]
    tot = 0.0
    for v in q.a_s:
        tot += v
    for v in bs:
        tot -= v
    q.b_s = bs + [tot if tot > 0 else 0.0]
    q.trials = n_trials(m, n)
    return "lp", q


def finish(q, xs):
    """path_decompose after the LP solves (path_decompose_sparse.py:118-192).
    xs: float64 [mn, trials] = scaled trial solutions, cell index j*m+i."""
    m, n, mn = q.m, q.n, q.m * q.n
    a, b, p, tol = q.a, q.b, q.p, q.tol
    thr = np.array([0.4 * min(a[i], b[j]) for j in range(n) for i in range(m)])
    temp_all = xs * q.scale
    kill = (temp_all < thr[:, None]) | (temp_all < tol) | (temp_all < 0)
    temp_all = np.where(kill, 0.0, temp_all)
    unsup = p > 0
    s_all = ((temp_all != 0) & unsup[:, None]).sum(axis=0)
    curr_min, curr_ans, curr_mult, curr_on = mn + 1, None, 0, 0.0
    for t in range(temp_all.shape[1]):
        temp = temp_all[:, t]
        s = int(s_all[t])
        if s > curr_min:
            continue
        dot = 0.0
        for k in range(mn):
            dot += p[k] * temp[k]
        if s < curr_min:
            curr_min, curr_ans, curr_mult, curr_on = s, temp, 0, dot
        else:
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
            if (abs(st - sc) < tol and dot < curr_on) or st > sc:
                curr_ans, curr_on = temp, dot
    answer = [[float(curr_ans[j * m + i]) for j in range(n)] for i in range(m)]
    non_unique = 1 if curr_mult > 1 else 0
    if q.sparsity and mn > q.sparsity:
        cells = [((i, j), answer[i][j]) for i in range(m) for j in range(n)]
        cells = sorted(cells, key=lambda c: c[1])[::-1][:q.sparsity]
        new = [[0.0] * n for _ in range(m)]
        for (i, j), v in cells:
            new[i][j] = v
        answer = new
    return answer, non_unique


def solve_batch(ctx, reqs, seed):
    """All trials of all pending decompositions in one device launch.  Returns [xs per request]."""
    if not reqs:
        return []
    np_ = len(reqs)
    m = np.array([q.m for q in reqs], dtype=np.uint32)
    n = np.array([q.n for q in reqs], dtype=np.uint32)
    tr = np.array([q.trials for q in reqs], dtype=np.uint32)
    pid = np.array([q.pid for q in reqs], dtype=np.uint64)
    ab = np.array([v for q in reqs for v in (q.a_s + q.b_s)], dtype=np.float64)
    mask = np.concatenate([(q.p > 0).astype(np.uint8) for q in reqs])
    sizes = (m.astype(np.int64) * n.astype(np.int64) * tr.astype(np.int64))
    out = np.empty(int(sizes.sum()), dtype=np.float64)
    _lib.check(_lib.lib().shn_lp_solve_batch(ctx.h, np_, m.ctypes.data, n.ctypes.data, tr.ctypes.data, pid.ctypes.data,
                                             ab.ctypes.data, mask.ctypes.data, C.c_uint64(seed), out.ctypes.data))
    res, off = [], 0
    for q, sz in zip(reqs, sizes.tolist()):
        res.append(out[off:off + sz].reshape(q.m * q.n, q.trials))
        off += sz
    return res


class _Node(object):
    __slots__ = ("string", "ine", "oute", "name", "weight", "L", "orig")

    def __init__(self, string, weight, L, name, orig):
        self.string, self.weight, self.L, self.name, self.orig = string, weight, L, name, orig
        self.ine, self.oute = [], []


def _key(n):
    return int(n.name.split("_")[0])


def component_coroutine(nodes, edges, paths, comp_id):
    """Generator: yields ('lp', _Req) and receives (answer, non_unique); returns the transcripts
    [(i, seq, avg_wt, node_names)] of one component (algorithm_SF.py:864-934)."""
    byid, allnodes = {}, []
    for nid, bases, cc, _norm in nodes:                       # ParseNodeFile :118-141
        x = _Node(bases, float(cc), len(bases), str(nid), True)
        byid[str(nid)] = x
        allnodes.append(x)
    for a, b, ov, cc, norm in edges:                          # ParseEdgeFile :143-160
        s, e = byid[str(a)], byid[str(b)]
        s.oute.append([e, int(ov), float(cc), float(norm)])
        e.ine.append([s, int(ov), float(cc), float(norm)])
    known, pfn = [], {id(x): [] for x in allnodes}            # ParseKnownPathsFile :91-115
    for i, p in enumerate(paths):
        lst = [byid[str(h)] for h in p]
        for x in lst:
            if len(pfn[id(x)]) < PATH_SPARSITY:
                pfn[id(x)].append(i)
        known.append(lst)
    S = _Node("Start_", 0, 0, "S", False)                     # findStartAndEnd2 :227-247
    E = _Node("_End", 0, 0, "E", False)
    for x in allnodes:
        if not x.ine:
            x.ine.append([S, 0, x.weight, 0])
            S.oute.append([x, 0, x.weight, 0])
            S.weight += float(x.weight)
        if not x.oute:
            x.oute.append([E, 0, x.weight, 0])
            E.ine.append([x, 0, x.weight, 0])
            E.weight += float(x.weight)
    allnodes += [S, E]

    def reducible():                                           # search() :357-370
        return [x for x in allnodes if x.ine and x.oute and x is not S and x is not E and len(x.ine) > 1]

    if len(allnodes) > 3:                                      # :878-883
        n_dec = 0
        while True:                                            # algorithm2 :373-561
            idx = 0
            while idx < len(allnodes):
                node = allnodes[idx]
                idx += 1
                if node is S or node is E or len(node.ine) <= 1:
                    continue
                if not node.oute:
                    node.oute.append([E, 0, node.weight, 0])
                    E.ine.append([node, 0, node.weight, 0])
                    E.weight += float(node.weight)
                inn = [e[0] for e in node.ine]
                outn = [e[0] for e in node.oute]
                a = [float(e[2]) for e in node.ine]
                b = [float(e[2]) for e in node.oute]
                in_attr = {id(e[0]): [e[1], e[3]] for e in node.ine}
                out_attr = {id(e[0]): [e[1], e[3]] for e in node.oute}
                m, n = len(a), len(b)
                P = [[0] * n for _ in range(m)]
                if node.orig and pfn[id(node)]:                # support matrix :438-498 (singleton constituents)
                    mine = set(pfn[id(node)])
                    for mi, u in enumerate(inn):
                        if not u.orig:
                            continue
                        for ni, w in enumerate(outn):
                            if not w.orig:
                                continue
                            for cp in mine & set(pfn[id(u)]) & set(pfn[id(w)]):
                                nl = known[cp]
                                k = next(q for q, x in enumerate(nl) if x is node)
                                lg = k == 0 or nl[k - 1].string == u.string
                                rg = k == len(nl) - 1 or nl[k + 1].string == w.string
                                if lg and rg:
                                    P[mi][ni] = 1
                kind, *rest = prepare(a, b, P, (comp_id << 20) + n_dec, PATH_SPARSITY)
                n_dec += 1
                if kind == "done":
                    flow = rest[0]
                else:
                    flow, _nu = yield rest[0]
                for i in range(m):
                    for j in range(n):
                        cc = flow[i][j]
                        if cc != 0:
                            ia, oa = in_attr[id(inn[i])], out_attr[id(outn[j])]
                            nn = _Node(node.string, cc, node.L, node.name + "_[" + str(i) + "," + str(j) + "]", False)
                            nn.ine.append([inn[i], ia[0], cc, ia[1]])
                            inn[i].oute.append([nn, ia[0], cc, ia[1]])
                            nn.oute.append([outn[j], oa[0], cc, oa[1]])
                            outn[j].ine.append([nn, oa[0], cc, oa[1]])
                            allnodes.append(nn)
                for e in node.ine:                              # :532-543 (remove while iterating, as written)
                    t = e[0]
                    for oe in t.oute:
                        if oe[0] is node:
                            t.oute.remove(oe)
                for e in node.oute:
                    t = e[0]
                    for ie in t.ine:
                        if ie[0] is node:
                            t.ine.remove(ie)
                k = next((q for q, x in enumerate(allnodes) if x is node), None)
                if k is not None:
                    del allnodes[k]                             # the list iterator then skips one element (quirk 21)
            if not reducible():
                break
            allnodes[:] = [x for x in allnodes if x is not S and x is not E]
            allnodes.sort(key=_key)
            allnodes.append(E)
            allnodes.insert(0, S)

    out = []                                                   # read_Y_paths :564-613

    def rec(node, s, names, overlap, sw, sn):
        cur = s + node.string[overlap:]
        cn = names + "->" + node.name.split("_")[0]
        if not node.oute:
            if cur[-4:] != "_End":
                return
            out.append([cur[:-4], float(sw) / sn if sn > 0 else 0, cn])
            return
        sw += node.weight
        sn += node.L
        for e in node.oute:
            rec(e[0], cur, cn, int(e[1]), sw, sn)

    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(max(lim, 100000))
    try:
        rec(S, "", "", 0, 0, 0)
    finally:
        sys.setrecursionlimit(lim)
    return [(i, s[6:], w, names) for i, (s, w, names) in enumerate(out) if len(s[6:])]


def sparse_flow_components(ctx, components, seed=0):
    """Run algorithm_SF on a list of components [(nodes, edges, paths)] concurrently: every round
    gathers one pending decomposition per component and solves all their LP trials in one device
    batch.  Component c uses RNG problem ids (c << 20) + call number.  Returns [transcripts]."""
    gens = [component_coroutine(nd, ed, pt, c) for c, (nd, ed, pt) in enumerate(components)]
    results = [None] * len(gens)
    pending = {}
    for c, g in enumerate(gens):
        try:
            pending[c] = next(g)
        except StopIteration as stop:
            results[c] = stop.value
    while pending:
        order = sorted(pending)
        xs = solve_batch(ctx, [pending[c] for c in order], seed)
        nxt = {}
        for c, x in zip(order, xs):
            ans = finish(pending[c], x)
            try:
                nxt[c] = gens[c].send(ans)
            except StopIteration as stop:
                results[c] = stop.value
        pending = nxt
    return results


def fasta_records(sname, comp, transcripts):
    """Header format of algorithm_SF.py:608-609."""
    return "".join(">Shannon_%s %s_%d\t%s\t%s\n%s\n" % (sname, comp, i, str(w), names, seq) for i, seq, w, names in transcripts)


def single_nodes_fasta(sname, single_rows):
    """single_nodes_to_fasta, algorithm_SF.py:74-88 -- including its quirk of not skipping the
    header line of single_nodes.txt (record 0 is `Copycount:Copycount` / `Bases`)."""
    rows = [("ID", "Bases", "Copycount", "Normalization")] + list(single_rows)
    return "".join(">Shannon_%s_single_%d\t Copycount:%s\n%s\n" % (sname, i, str(cc), b) for i, (_h, b, cc, _n) in enumerate(rows))
