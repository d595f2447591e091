"""Partitioning + per-partition k1-mer / read routing -- host mirror of the reference's
kmers_for_component.py (rows a8-a11) over the HIP routing kernel (csrc/route.hip).

gpmetis (METIS 5, external, unpinned) is replaced by `partition_graph` below: a deterministic
multilevel k-way partitioner (heavy-edge-matching coarsening, greedy growing on the coarsest graph,
boundary refinement on every level) honouring the same balance bound (ufactor).  Its output is not
METIS's -- partitioned configurations are parity-checked given the same partition vector.
"""
import ctypes as C
import math
import numpy as np
from . import _lib, device
from .extension_correction import windows_to_keys, windows_to_keys_many


def n_partitions(num_contigs, partition_size):
    """kmers_for_component.py:217."""
    return min(int(math.ceil(float(num_contigs) / float(partition_size))), 100)


def parse_metis(text):
    lines = text.splitlines()
    n = int(lines[0].split()[0])
    adj = []
    for i in range(n):
        tok = lines[1 + i].split() if 1 + i < len(lines) else []
        adj.append([(int(tok[j]) - 1, int(tok[j + 1])) for j in range(0, len(tok), 2)])
    return adj


def partition_graph(metis_text, n_parts, ufactor=1000):
    """Deterministic stand-in for `gpmetis -ufactor=U graph P` (kmers_for_component.py:221,234), built like METIS itself -- a
    multilevel k-way scheme: (1) coarsening by heavy-edge matching (vertices in index order, each unmatched vertex pairs with the
    unmatched neighbour behind its heaviest edge; matched pairs become one vertex, parallel edges add up) until the graph is small;
    (2) an initial partition of the coarsest graph by greedy graph growing on vertex weights; (3) projection back level by level,
    with a k-way boundary refinement on every level.  Balance: every part holds at most (1 + U/1000) * n / P vertices.  The output
    is not METIS's digit for digit (gpmetis is external, randomised and unpinned: parity cases replay a given partition vector);
    on planted partitions it ends within a few per cent of the planted cut (tests/test_host_graph.py).
    Runs in the library (shn_partition_metis, csrc/partition_host.hip); partition_graph_py is the same algorithm statement for
    statement in Python (the readable form; tests compare the two)."""
    text = metis_text.encode() if isinstance(metis_text, str) else bytes(metis_text)
    n = int(text.split(None, 1)[0]) if text.strip() else 0
    out = np.zeros(max(n, 1), dtype=np.int32)
    _lib.check(_lib.lib().shn_partition_metis(text, len(text), n, int(n_parts), int(ufactor), out.ctypes.data))
    return out[:n].tolist()


def partition_graph_py(metis_text, n_parts, ufactor=1000):
    """partition_graph in Python (the readable mirror of csrc/partition_host.hip)"""
    adj = parse_metis(metis_text)
    return multilevel_partition(adj, n_parts, ufactor)


def _grow_partition(adj, vw, n_parts):
    """greedy graph growing on vertex weights: parts one at a time from the lowest-numbered free vertex, always absorbing the free
    vertex with the largest total edge weight into the growing part (ties: lowest index), up to ceil(W / P) weight; leftovers go
    to the lightest part"""
    n = len(adj)
    part = [-1] * n
    total = sum(vw)
    target = int(math.ceil(total / float(n_parts)))
    nxt = 0
    sizes = [0] * n_parts
    for p in range(n_parts):
        while nxt < n and part[nxt] != -1:
            nxt += 1
        if nxt >= n:
            break
        gain = {nxt: 0}
        while sizes[p] < target and gain:
            v = min(gain, key=lambda x: (-gain[x], x))
            del gain[v]
            part[v] = p
            sizes[p] += vw[v]
            for u, w in adj[v]:
                if part[u] == -1:
                    gain[u] = gain.get(u, 0) + w
            if not gain and sizes[p] < target:
                while nxt < n and part[nxt] != -1:
                    nxt += 1
                if nxt < n:
                    gain[nxt] = 0
    for v in range(n):
        if part[v] == -1:
            p = min(range(n_parts), key=lambda q: (sizes[q], q))
            part[v] = p
            sizes[p] += vw[v]
    return part


def multilevel_partition(adj, n_parts, ufactor=1000, coarse_min=None):
    n = len(adj)
    if n_parts <= 1 or n == 0:
        return [0] * n
    levels = []                                   # (adjacency, vertex weights, map to the next coarser level)
    cur, vw = adj, [1] * n
    stop = coarse_min if coarse_min is not None else max(20 * n_parts, 64)
    max_vw = max(1, int(1.5 * n / float(stop)))   # (no coarse vertex heavier than that: the initial partition must still balance)
    while len(cur) > stop:
        cadj, cvw, cmap = _coarsen_bounded(cur, vw, max_vw)
        if len(cadj) > 0.95 * len(cur):           # (nothing left to match)
            break
        levels.append((cur, vw, cmap))
        cur, vw = cadj, cvw
    part = _grow_partition(cur, vw, n_parts)
    part = refine_partition(cur, part, n_parts, ufactor, vw=vw, total=n)
    for fine_adj, fine_vw, cmap in reversed(levels):
        part = [part[cmap[v]] for v in range(len(fine_adj))]
        part = refine_partition(fine_adj, part, n_parts, ufactor, vw=fine_vw, total=n)
    # (a part the refinement emptied -- possible on tiny graphs -- takes the lightest boundary vertex of the largest part back)
    sizes = [0] * n_parts
    for p in part:
        sizes[p] += 1
    for q in range(n_parts):
        if sizes[q] == 0:
            big = max(range(n_parts), key=lambda x: (sizes[x], -x))
            v = next(i for i in range(n) if part[i] == big)
            part[v] = q
            sizes[big] -= 1
            sizes[q] += 1
    return part


def _coarsen_bounded(adj, vw, max_vw):
    """_coarsen, with pairs heavier than max_vw left unmatched"""
    n = len(adj)
    match = [-1] * n
    cmap = [-1] * n
    nc = 0
    for v in range(n):
        if match[v] != -1:
            continue
        best, bw = -1, -1
        for u, w in adj[v]:
            if u != v and match[u] == -1 and vw[u] + vw[v] <= max_vw and (w > bw or (w == bw and u < best)):
                best, bw = u, w
        match[v] = best if best >= 0 else v
        cmap[v] = nc
        if best >= 0:
            match[best] = v
            cmap[best] = nc
        nc += 1
    cvw = [0] * nc
    cadj = [dict() for _ in range(nc)]
    for v in range(n):
        cv = cmap[v]
        cvw[cv] += vw[v]
        d = cadj[cv]
        for u, w in adj[v]:
            cu = cmap[u]
            if cu != cv:
                d[cu] = d.get(cu, 0) + w
    return [sorted(d.items()) for d in cadj], cvw, cmap


def edge_cut(adj, part):
    """total weight of the edges between different parts (what gpmetis minimises)"""
    return sum(w for v, nb in enumerate(adj) for u, w in nb if u > v and part[u] != part[v])


def refine_partition(adj, part, n_parts, ufactor=1000, max_passes=16, vw=None, total=None):
    """k-way boundary refinement of a partition (the uncoarsening phase of the multilevel scheme gpmetis runs): vertices in index
    order, a vertex moves to the neighbouring part it is connected to most strongly if that lowers the cut, the target stays within
    the balance bound (1 + U/1000) * n / P (in vertex weight: vw, on the coarse levels) and its own part does not become empty;
    passes until nothing moves.  Every move lowers the cut, so it ends; deterministic."""
    n = len(adj)
    part = list(part)
    if vw is None:
        vw = [1] * n
    if total is None:
        total = sum(vw)
    sizes = [0] * n_parts
    cnt = [0] * n_parts
    for v, p in enumerate(part):
        sizes[p] += vw[v]
        cnt[p] += 1
    max_size = max(int(math.ceil(total / float(n_parts))), int((1.0 + ufactor / 1000.0) * total / float(n_parts)))
    for _ in range(max_passes):
        moved = 0
        for v in range(n):
            pv = part[v]
            if cnt[pv] <= 1 or not adj[v]:
                continue
            conn = {}
            for u, w in adj[v]:
                if u != v:
                    conn[part[u]] = conn.get(part[u], 0) + w
            own = conn.get(pv, 0)
            best, bw = -1, own
            for q in sorted(conn):
                if q != pv and conn[q] > bw and sizes[q] + vw[v] <= max_size:
                    best, bw = q, conn[q]
            if best >= 0:
                part[v] = best
                sizes[pv] -= vw[v]
                sizes[best] += vw[v]
                cnt[pv] -= 1
                cnt[best] += 1
                moved += 1
        if not moved:
            break
    return part


def weight_updated_graph(metis_text, part, penalty=5):
    """weight_updated_graph.py:24-42: multiply weights of edges cut by `part` by `penalty` (shn_metis_reweight; the Python form:
    weight_updated_graph_py)."""
    text = metis_text.encode() if isinstance(metis_text, str) else bytes(metis_text)
    pv = np.ascontiguousarray(part, dtype=np.int32)
    n = C.c_uint64(0)
    L = _lib.lib()
    # one call: a weight times the penalty grows by as many digits as the penalty has, a line may gain its missing newline
    grow = len(str(int(penalty))) if int(penalty) > 0 else 24
    out = np.empty(len(text) + grow * (text.count(b"\t") // 2 + 1) + len(pv) + 64, dtype=np.uint8)
    try:
        _lib.check(L.shn_metis_reweight(text, len(text), pv.ctypes.data, len(pv), int(penalty), out.ctypes.data, len(out), C.byref(n)))
    except _lib.ShannonError as ex:
        if "too small" not in str(ex):
            raise
        out = np.empty(max(1, n.value), dtype=np.uint8)                  # (texts with spaces between the numbers, signs, ...: the size is known now)
        _lib.check(L.shn_metis_reweight(text, len(text), pv.ctypes.data, len(pv), int(penalty), out.ctypes.data, len(out), C.byref(n)))
    return out[:n.value].tobytes().decode()


def weight_updated_graph_py(metis_text, part, penalty=5):
    """weight_updated_graph.py:24-42 in Python."""
    lines = metis_text.splitlines()
    out = [lines[0] + "\n"]
    for i, line in enumerate(lines[1:], start=1):
        tok = line.split()
        new = ""
        for j in range(0, len(tok), 2):
            nb = tok[j]
            cut = int(part[i - 1]) != int(part[int(nb) - 1])
            new += nb + "\t" + (str(penalty * int(tok[j + 1])) if cut else tok[j + 1]) + "\t"
        out.append(new + "\n")
    return "".join(out)


def _contigs_by_part(out, prefix, part, contigs):
    """out[prefix + str(p)] gets the contigs of part p in their order, the parts in the order of their first contig -- what the loop
    `for j, p in enumerate(part): out.setdefault(prefix + str(p), []).append(contigs[j])` of kmers_for_component.py:239-262 leaves
    (grouped with one stable sort: 240,000 formatted dictionary look-ups per step at bench.py --config 2p were 0.1 s)"""
    pv = np.asarray(part, dtype=np.int64)
    if not len(pv):
        return
    order = np.argsort(pv, kind="stable")
    cuts = np.flatnonzero(np.diff(pv[order])) + 1
    groups = np.split(order, cuts)
    for gi in np.argsort(np.asarray([g[0] for g in groups]), kind="stable").tolist():
        g = groups[gi].tolist()
        out.setdefault(prefix + str(int(pv[g[0]])), []).extend([contigs[j] for j in g])


def build_partitions(res, K, partition_size=500, overload=2, penalty=5, repartition=True, part_vectors=None):
    """kmers_for_component.py:207-305.  `res`: ExtensionResult.  part_vectors: optional
    [(part, part_r2)] per big component (to replay a given gpmetis output).  Returns
    (new_components {name: [contig]}, components_broken {i: P})."""
    ufactor = int(1000.0 * overload - 1000.0)
    new_components, broken = {}, {}

    def both_runs(i):
        # gpmetis, then gpmetis again on the graph with the cut edges' weights multiplied (:207-237); one component's two runs depend on
        # one another, the components do not: they run side by side (the native partitioner releases the GIL)
        contigs, metis = res.big_components[i]
        P = n_partitions(len(contigs), partition_size)
        p1 = partition_graph(metis, P, ufactor)
        return p1, (partition_graph(weight_updated_graph(metis, p1, penalty), P, ufactor) if repartition else None)
    computed = None
    if part_vectors is None and len(res.big_components) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(res.big_components))) as ex:
            computed = list(ex.map(both_runs, range(len(res.big_components))))
    for i, (contigs, metis) in enumerate(res.big_components):
        P = n_partitions(len(contigs), partition_size)
        broken[i] = P
        if part_vectors is not None:
            p1, p2 = part_vectors[i]
        else:
            p1, p2 = computed[i] if computed is not None else both_runs(i)
        _contigs_by_part(new_components, "c%d_" % (i + 1), p1, contigs)
        if repartition and p2 is not None:
            _contigs_by_part(new_components, "r2_c%d_" % (i + 1), p2, contigs)
    for i, contigs in enumerate(res.remaining):
        for c in contigs:
            new_components.setdefault("cremaining%d" % (i + 1), []).append(c)
    return new_components, broken


class Routes(object):
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def download(self):
        n = int(_lib.lib().shn_routes_size(self.h))
        pid = np.empty(n, np.uint32)
        ridx = np.empty(n, np.uint32)
        _lib.check(_lib.lib().shn_routes_download(self.ctx.h, self.h, pid.ctypes.data, ridx.ctypes.data))
        return pid, ridx

    def bounds(self, n_parts, split):
        start = np.zeros(n_parts + 1, np.uint64)
        below = np.zeros(max(n_parts, 1), np.uint64)
        _lib.check(_lib.lib().shn_routes_bounds(self.ctx.h, self.h, int(n_parts), int(split), start.ctypes.data, below.ctypes.data))
        return start.astype(np.int64), below[:n_parts].astype(np.int64)

    def download_range(self, lo, n):
        out = np.empty(int(n), np.uint32)
        _lib.check(_lib.lib().shn_routes_download_range(self.ctx.h, self.h, int(lo), int(n), out.ctypes.data))
        return out

    def close(self):
        if self.h:
            _lib.lib().shn_routes_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RouteView(object):
    """The routes of one partition left on the device: length, forward-half count and slices on demand (a partition's
    graph consumes only a capped prefix of each half; the routes of a highly expressed contig run into the millions)."""

    def __init__(self, routes, lo, n, below):
        self.routes, self.lo, self.n, self.below = routes, int(lo), int(n), int(below)

    def __len__(self):
        return self.n

    def count_below_split(self):
        return self.below

    def __getitem__(self, sl):
        if not (isinstance(sl, slice) and sl.step in (None, 1)):          # (anything but a plain slice: the whole list, then numpy's indexing)
            return self.routes.download_range(self.lo, self.n)[sl]
        a, b, _ = sl.indices(self.n)
        return self.routes.download_range(self.lo + a, max(0, b - a))

    def __array__(self, dtype=None, copy=None):
        a = self.routes.download_range(self.lo, self.n)
        return a if dtype is None else a.astype(dtype)


def make_table(ctx, keys, values, k, canonical=False):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    h = C.c_void_p()
    _lib.check(_lib.lib().shn_table_create(ctx.h, keys.ctypes.data, values.ctypes.data, len(keys), k, 1 if canonical else 0, C.byref(h)))
    return device.Table(ctx, h)


def kmers_for_component(ctx, res, reads1, reads2, K, partition_size=500, overload=2, penalty=5, repartition=True,
                        part_vectors=None, want_rows=True, timings=None, lazy_routes=False, lazy_graph_inputs=False, strand_specific=False):
    """Rows a8-a11.  reads1/reads2: device.Reads (reads2 None for single-end).  strand_specific (-s / --ss, shannon.py:407-411):
    the read files are reads / (reads_1, RC(reads_2)), not strand-doubled -- routes hold plain read indices.  Returns dict with
      new_components {name: [contig]}          (kmers_for_component.py:244-305)
      k1mers {name: [(k1mer, weight)]}         (:452-477, == component{name}k1mers_allowed.dict)
      contig_weights {name: [[w..] per contig]} (--inMem form :468-469)
      routes {name: uint32 array of doubled read indices in input order}   (:322-403)
    """
    import time as _t
    _t0 = [_t.time()]

    def lap(name):
        if timings is not None:
            now = _t.time()
            timings[name] = timings.get(name, 0.0) + now - _t0[0]
            _t0[0] = now

    k1 = K + 1
    comps, broken = build_partitions(res, K, partition_size, overload, penalty, repartition, part_vectors)
    names = list(comps)
    lap("route.partitions")
    pid_of = {n: i for i, n in enumerate(names)}
    files, cw = {}, {}
    import os
    # (the loops over all contigs below run inside C -- map / chain / fromiter: at bench.py --config 2p every contig is in two
    # partitions, 240,000 list entries, and generator expressions over them were 0.1 s of this stage)
    import itertools
    flat = list(itertools.chain.from_iterable(comps[name] for name in names))
    flat_len = np.fromiter(map(len, flat), dtype=np.int64, count=len(flat))
    n_windows = int(np.maximum(flat_len - k1 + 1, 0).sum())
    _pg = os.environ.get("SHN_PROBE_GPU", "")
    probe_h = None
    flat_text = None
    if k1 <= 32 and (_pg == "1" or (_pg != "0" and n_windows >= (1 << 20))):
        # k1mers2component on the GPU (shn_probe_build): one device sort of the k1-windows of all partition contigs
        text = off = None
        raw = getattr(res, "contig_raw", None)
        if raw is not None and flat and len(getattr(res, "contigs", ())) == len(raw[2]):
            # the contigs are the string objects of res.contigs (the partition lists hold references): their bytes are gathered from
            # the candidate buffer of the contig stage on host threads instead of joining and encoding the strings
            where = dict(zip(map(id, res.contigs), range(len(res.contigs))))
            try:
                order = raw[2][np.fromiter(map(where.__getitem__, map(id, flat)), dtype=np.int64, count=len(flat))]
                text, off = _lib.gather_segments(raw[0], raw[1], order, threads=_lib.host_cpus())
            except KeyError:
                text = off = None
        if text is None:
            text = np.frombuffer("".join(flat).encode(), dtype=np.uint8) if flat else np.zeros(1, np.uint8)
            off = np.zeros(len(flat) + 1, dtype=np.uint64)
            if flat:
                off[1:] = np.cumsum(flat_len, dtype=np.uint64)
        part_of = np.ascontiguousarray(np.repeat(np.arange(len(names), dtype=np.uint32), [len(comps[name]) for name in names]), dtype=np.uint32) \
            if flat else np.zeros(1, np.uint32)
        flat_text = (text, off, part_of, len(flat))          # (the unitig batch of the graph stage takes the same contigs in the same order)
        probe_h = C.c_void_p()
        _lib.check(_lib.lib().shn_probe_build(ctx.h, text.ctypes.data, off.ctypes.data, len(flat), part_of.ctypes.data, len(names), int(k1),
                                              C.byref(probe_h)))
        n_sets = int(_lib.lib().shn_probe_n_sets(probe_h))
        set_off = np.zeros(n_sets + 1, dtype=np.uint32)
        set_mem = np.zeros(max(int(_lib.lib().shn_probe_n_members(probe_h)), 1), dtype=np.uint32)
        _lib.check(_lib.lib().shn_probe_sets(probe_h, set_off.ctypes.data, set_mem.ctypes.data))
        lap("route.k1mer map")
        h = C.c_void_p()
        try:
            _lib.check(_lib.lib().shn_route_reads_mode(ctx.h, reads1.h, reads2.h if reads2 is not None else None, k1,
                                                       C.c_void_p(_lib.lib().shn_probe_table(probe_h)), set_off.ctypes.data, set_mem.ctypes.data, n_sets,
                                                       1 if strand_specific else 0, C.byref(h)))
        finally:
            _lib.lib().shn_probe_destroy(probe_h)
        lap("route.kernel")
    if probe_h is None:
        h = _host_probe_and_route(ctx, comps, names, pid_of, reads1, reads2, k1, lap, strand_specific)
    routes = Routes(ctx, h)
    if lazy_routes:                                   # routes stay on the device; RouteView fetches what is asked for
        start, below = routes.bounds(len(names), len(reads1))
        by_part = {n: RouteView(routes, start[i], start[i + 1] - start[i], below[i]) for i, n in enumerate(names)}
    else:
        # (the partition column is not fetched: the routes are sorted by partition and the device knows where each one starts)
        start, _below = routes.bounds(len(names), len(reads1))
        ridx = routes.download_range(0, int(start[len(names)]))
        by_part = {n: ridx[int(start[i]):int(start[i + 1])] for i, n in enumerate(names)}
    lap("route.download")
    out = _finish_partitions(res, comps, broken, names, by_part, files, cw, k1, K, want_rows, lazy_graph_inputs, lap)
    out["flat_text"] = flat_text
    # the routes also stay where the routing left them (1 GB at BASELINE configs[2]; freed with this dict): the graph stage's
    # duplicate search reads a partition's list in place (shn_mbgraph_run_routes) -- {name: first entry of the partition}
    out["routes_dev"] = (routes, {n: int(start[i]) for i, n in enumerate(names)})
    return out


def _host_probe_and_route(ctx, comps, names, pid_of, reads1, reads2, k1, lap, strand_specific=False):
    """k1mers2component with numpy (small inputs; the readable form of csrc/probe_gpu.hip) + the routing kernel; returns the
    shn_routes handle"""
    allk, allp = [], []
    for name in names:
        ks, _nw = windows_to_keys_many(comps[name], k1)
        allk.append(ks)
        allp.append(np.full(len(ks), pid_of[name], dtype=np.uint32))
    if allk:
        allk = np.concatenate(allk)
        allp = np.concatenate(allp)
    else:
        allk, allp = np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    # distinct (k1-mer, partition) pairs, grouped by k1-mer; a k1-mer lies on exactly one contig, so its set
    # has one partition (remaining bin) or two (gpmetis run + r2 run): singleton sets are handled vectorised
    pbits = max(1, int(len(names)).bit_length())
    if 2 * k1 + pbits <= 64:
        # one sort of (k1-mer << pbits | partition) instead of an indirect two-key sort
        code = np.sort((allk << np.uint64(pbits)) | allp.astype(np.uint64))
        if len(code):
            code = code[np.concatenate([[True], code[1:] != code[:-1]])]
        sk, sp = code >> np.uint64(pbits), (code & np.uint64((1 << pbits) - 1)).astype(np.uint32)
    else:
        order = np.lexsort((allp, allk))
        sk, sp = allk[order], allp[order]
        if len(sk):
            keepm = np.ones(len(sk), dtype=bool)
            keepm[1:] = (sk[1:] != sk[:-1]) | (sp[1:] != sp[:-1])
            sk, sp = sk[keepm], sp[keepm]
    # sk is sorted: group boundaries instead of np.unique's second sort
    if len(sk):
        first = np.concatenate([[True], sk[1:] != sk[:-1]])
        start = np.nonzero(first)[0]
        uk = sk[start]
        cntk = np.diff(np.concatenate([start, [len(sk)]]))
    else:
        uk, start, cntk = np.zeros(0, np.uint64), np.zeros(0, np.int64), np.zeros(0, np.int64)
    set_ids, sets, set_index = np.zeros(len(uk), np.uint32), [], {}
    single = cntk == 1
    if single.any():
        pids = sp[start[single]]
        upid, inv = np.unique(pids, return_inverse=True)
        base = len(sets)
        for pv in upid.tolist():
            set_index[(pv,)] = len(sets)
            sets.append((pv,))
        set_ids[single] = (base + inv + 1).astype(np.uint32)
    double = cntk == 2                               # gpmetis run + r2 run: the common multi-partition case, vectorised
    if double.any():
        p0, p1 = sp[start[double]].astype(np.int64), sp[start[double] + 1].astype(np.int64)
        code = p0 * (len(names) + 1) + p1
        ucode, inv = np.unique(code, return_inverse=True)
        base = len(sets)
        for cv in ucode.tolist():
            tpl = (cv // (len(names) + 1), cv % (len(names) + 1))
            set_index[tpl] = len(sets)
            sets.append(tpl)
        set_ids[double] = (base + inv + 1).astype(np.uint32)
    for i in np.nonzero(cntk > 2)[0].tolist():
        tpl = tuple(sp[start[i]:start[i] + cntk[i]].tolist())
        sid = set_index.get(tpl)
        if sid is None:
            sid = set_index[tpl] = len(sets)
            sets.append(tpl)
        set_ids[i] = sid + 1
    set_off = np.zeros(len(sets) + 1, dtype=np.uint32)
    set_off[1:] = np.cumsum([len(s) for s in sets])
    set_mem = np.array([p for s in sets for p in s], dtype=np.uint32) if sets else np.zeros(1, np.uint32)
    lap("route.k1mer map")
    probe = make_table(ctx, uk, set_ids, k1, canonical=False)
    h = C.c_void_p()
    try:
        _lib.check(_lib.lib().shn_route_reads_mode(ctx.h, reads1.h, reads2.h if reads2 is not None else None, k1, probe.h,
                                                   set_off.ctypes.data, set_mem.ctypes.data, len(sets), 1 if strand_specific else 0, C.byref(h)))
    finally:
        probe.close()
    lap("route.kernel")
    return h


def _finish_partitions(res, comps, broken, names, by_part, files, cw, k1, K, want_rows, lazy_graph_inputs, lap):
    # per-partition k1-mer rows with weights from the allowed dict (:452-477)
    rows_bytes, n_nodes, n_rows_of = {}, {}, {}
    for name in names:
        cl = comps[name]
        n_rows_of[name] = (int(np.maximum(np.fromiter(map(len, cl), dtype=np.int64, count=len(cl)) - (k1 - 1), 0).sum()) if cl else 0)
        if lazy_graph_inputs and not want_rows:
            # the graph stage takes its K-mer graph from the GPU unitig builder (which also counts the distinct K-mers);
            # the byte rows are made on demand (a partition with a cycle of condensable edges, SHN_GRAPH_CHECK)
            rows_bytes[name] = (lambda nm=name: _rows_bytes(comps[nm], k1))
            continue
        # fixed-width byte form of the k1-mer file (what the native graph stage consumes) + #distinct K-mers
        rb = None
        if comps[name]:
            try:                                     # native pass over the joined contigs (shn_string_windows)
                _k, rb, _nw = _lib.string_windows(comps[name], k1, want_keys=False, want_rows=True)
            except _lib.ShannonError:
                rb = None
        if rb is None:
            chunks = []
            for contig in comps[name]:
                b = np.frombuffer(contig.encode(), dtype=np.uint8)
                nwin = len(b) - k1 + 1
                if nwin > 0:
                    chunks.append(np.lib.stride_tricks.sliding_window_view(b, k1).reshape(-1))
            rb = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
        rows_bytes[name] = np.ascontiguousarray(rb)
        kk, _nw = windows_to_keys_many(comps[name], K)
        kk = np.sort(kk)
        n_nodes[name] = int((kk[1:] != kk[:-1]).sum()) + 1 if len(kk) else 0
        if want_rows:
            rows, ws = [], []
            for contig in comps[name]:
                wl = []
                for i in range(len(contig) - k1 + 1):
                    km = contig[i:i + k1]
                    w = res.allowed.get(km, 0)
                    wl.append(w)
                    rows.append((km, w))
                ws.append(wl)
            files[name] = rows
            cw[name] = ws
    lap("route.rows")
    return {"new_components": comps, "components_broken": broken, "k1mers": files, "contig_weights": cw, "routes": by_part,
            "k1mer_bytes": rows_bytes, "n_kmer_nodes": n_nodes, "n_k1mer_rows": n_rows_of}


def _rows_bytes(contigs, k1):
    """fixed-width byte form of a partition's k1-mer file: every k1-window of every contig, in order"""
    if not contigs:
        return np.zeros(0, np.uint8)
    try:
        _k, rb, _nw = _lib.string_windows(contigs, k1, want_keys=False, want_rows=True)
        return np.ascontiguousarray(rb)
    except _lib.ShannonError:
        chunks = []
        for contig in contigs:
            b = np.frombuffer(contig.encode(), dtype=np.uint8)
            if len(b) - k1 + 1 > 0:
                chunks.append(np.lib.stride_tricks.sliding_window_view(b, k1).reshape(-1))
        return np.ascontiguousarray(np.concatenate(chunks)) if chunks else np.zeros(0, np.uint8)


class ReadStore(object):
    """Host copy of the input reads, addressed by the doubled index of the routing kernel."""

    def __init__(self, r1, r2=None):
        self.r1, self.r2 = r1, r2         # lists of strings or uint8 code matrices
        self.n = len(r1)
        # Row buffers handed to the graph stage are recycled (release()): a partition at the read cap gathers 400 MB of rows, a run
        # at 100 M reads 24 GB per step, and with 64 partition threads faulting fresh pages in at the same time the kernel's
        # address-space lock, not the work, set the pace of the stage.
        import threading
        self._pool, self._pool_lock, self._off = [], threading.Lock(), None

    def _rows_buffer(self, n_rows, L):
        need = n_rows * L
        with self._pool_lock:
            best = -1
            for i, b in enumerate(self._pool):
                if len(b) >= need and (best < 0 or len(b) < len(self._pool[best])):
                    best = i
            buf = self._pool.pop(best) if best >= 0 else None
        if buf is None:
            buf = np.empty(max(need, 1 << 20), dtype=np.uint8)
            try:                                     # transparent huge pages where the system offers them on request (madvise mode)
                import ctypes as _C
                _C.CDLL(None).madvise(_C.c_void_p(buf.ctypes.data & ~4095), _C.c_size_t(buf.nbytes), 14)
            except Exception:
                pass
        return buf

    def release(self, flat_rows):
        """give a buffer of gather_codes back (the flat array it returned, or its base)"""
        b = flat_rows
        while getattr(b, "base", None) is not None and isinstance(b.base, np.ndarray):
            b = b.base
        if isinstance(b, np.ndarray) and b.dtype == np.uint8 and b.ndim == 1:
            with self._pool_lock:
                if len(self._pool) < 256:
                    self._pool.append(b)

    def _offsets(self, n_rows, L):
        o = self._off
        if o is None or len(o) < n_rows + 1 or o[1] != L:
            o = np.arange(max(n_rows + 1, 1 << 16), dtype=np.uint64) * np.uint64(L)
            self._off = o
        return o[:n_rows + 1]

    @staticmethod
    def _get(src, i):
        s = src[i]
        if isinstance(s, str):
            return s
        return device.ALPHA_BYTES[np.minimum(s, 4)].tobytes().decode()

    @property
    def ragged(self):
        """the reads are device.RaggedCodes (reads of different lengths from shn_reads_ingest_ragged)"""
        return isinstance(self.r1, device.RaggedCodes)

    def _gather_ragged(self, idx, mate, ss):
        """gather_codes / gather_codes_ss over RaggedCodes: the reads' codes as stored, one after the other, + offsets + rc flags"""
        idx = np.asarray(idx, dtype=np.int64)
        n = self.n
        if ss:
            src = self.r1 if mate == 1 else self.r2
            buf, off = src.take(idx)
            return buf, off, np.full(len(idx), 1 if mate == 2 else 0, dtype=np.uint8), 1
        second = idx >= n
        src2 = self.r1 if self.r2 is None else self.r2
        if len(idx) < 2 or bool((idx[1:] >= idx[:-1]).all()):          # ascending (how the routes come): forward half, then RC half
            f = int(np.searchsorted(idx, n))
            b1, o1 = self.r1.take(idx[:f])
            b2, o2 = src2.take(idx[f:] - n)
            buf = np.concatenate([b1[:int(o1[-1])], b2[:int(o2[-1])]]) if (int(o1[-1]) + int(o2[-1])) else np.zeros(1, np.uint8)
            off = np.concatenate([o1, o2[1:] + o1[-1]])
        else:
            parts = [(src2 if s2 else self.r1)[int(d - n if s2 else d)] for d, s2 in zip(idx.tolist(), second.tolist())]
            off = np.zeros(len(parts) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(x) for x in parts], dtype=np.uint64)
            buf = np.concatenate(parts) if parts else np.zeros(1, np.uint8)
        rc = second if (self.r2 is None or mate == 1) else ~second
        return np.ascontiguousarray(buf), off, np.ascontiguousarray(rc, dtype=np.uint8), 1

    @staticmethod
    def _rc(s):
        return s[::-1].translate(_RC)

    def gather_codes_ss(self, idx, mate):
        """gather_codes for a strand-specific run (-s, shannon.py:407-411): `idx` are plain read indices; mate 1 = reads_1[i] as
        stored, mate 2 = RC(reads_2[i]) (rows as stored + the rc flag set)."""
        idx = np.asarray(idx, dtype=np.int64)
        src = self.r1 if mate == 1 else self.r2
        if self.ragged and len(idx):
            return self._gather_ragged(idx, mate, True)
        if len(idx) == 0 or isinstance(src[0], str):
            seqs = [self._get(src, int(i)) for i in idx]
            if mate == 2:
                seqs = [self._rc(q) for q in seqs]
            off = np.zeros(len(seqs) + 1, dtype=np.uint64)
            if seqs:
                off[1:] = np.cumsum([len(q) for q in seqs], dtype=np.uint64)
            buf = np.frombuffer("".join(seqs).encode(), dtype=np.uint8) if seqs else np.zeros(1, np.uint8)
            return buf, off, None, 0
        rows = np.ascontiguousarray(src[idx])
        L = rows.shape[1]
        rc = np.full(len(idx), 1 if mate == 2 else 0, dtype=np.uint8)
        return (rows.reshape(-1), np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(L), rc, 1)

    def gather_codes(self, idx, mate):
        """(codes buffer, offsets, rc flags, enc) of reads `mate` of the doubled indices -- rows are gathered as
        stored; the reverse complement is left to the consumer (shn_mbgraph_run's rc flags)."""
        idx = np.asarray(idx, dtype=np.int64)
        n = self.n
        if self.ragged and len(idx):
            return self._gather_ragged(idx, mate, False)
        if len(idx) == 0 or isinstance(self.r1[0], str):
            b, o = self.gather(idx, mate)
            return b, o, None, 0
        second = idx >= n
        # one gather per source matrix straight into the output (idx is ascending: forward half first)
        L0 = self.r1.shape[1]
        rows = self._rows_buffer(len(idx), L0)[:len(idx) * L0].reshape(len(idx), L0)
        src2 = self.r1 if self.r2 is None else self.r2
        f = int(np.searchsorted(idx, n)) if (len(idx) < 2 or bool((idx[1:] >= idx[:-1]).all())) else -1
        if f >= 0 and len(idx) >= 4096 and self.r1.flags["C_CONTIGUOUS"] and src2.flags["C_CONTIGUOUS"] and self.r1.dtype == np.uint8:
            from . import _lib                                           # ascending: forward half, then RC half -- two threaded gathers
            _lib.gather_rows(self.r1, idx[:f], rows[:f])
            _lib.gather_rows(src2, idx[f:] - n, rows[f:])
        else:
            fwd, bwd = np.nonzero(~second)[0], np.nonzero(second)[0]
            if len(fwd):
                rows[fwd] = self.r1[idx[fwd]]
            if len(bwd):
                rows[bwd] = src2[idx[bwd] - n]
        if self.r2 is None:
            rc = second                                                     # SE: R[d] / RC(R[d-n])
        elif mate == 1:
            rc = second                                                     # R1[d] / RC(R2[d-n])
        else:
            rc = ~second                                                    # RC(R1[d]) / R2[d-n]
        L = rows.shape[1]
        return (rows.reshape(-1), self._offsets(len(idx), L), np.ascontiguousarray(rc, dtype=np.uint8), 1)

    def gather(self, idx, mate):
        """Reads `mate` (1 or 2) of the doubled indices `idx` as (uint8 ASCII buffer, uint64 offsets) -- the
        byte layout shn_mbgraph_run takes.  Vectorised for code matrices."""
        idx = np.asarray(idx, dtype=np.int64)
        n = self.n
        if len(idx) and not isinstance(self.r1[0], str) and not self.ragged:
            second = idx >= n
            i = np.where(second, idx - n, idx)
            if mate == 1:
                src_fwd = self.r1[i]                                    # d < n: R1[d]
                other = (self.r2 if self.r2 is not None else self.r1)[i]  # d >= n: RC(R2[d-n]) (SE: RC(R[d-n]))
                out = np.where(second[:, None], 3 - other[:, ::-1], src_fwd)
            else:
                out = np.where(second[:, None], self.r2[i], 3 - self.r1[i][:, ::-1])
            L = out.shape[1]
            buf = device.ALPHA_BYTES[out.astype(np.uint8)].reshape(-1)
            return np.ascontiguousarray(buf), (np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(L))
        strs = [self.mate1(int(d)) if mate == 1 else self.mate2(int(d)) for d in idx]
        joined = "".join(strs).encode()
        off = np.zeros(len(strs) + 1, dtype=np.uint64)
        if strs:
            off[1:] = np.cumsum([len(x) for x in strs], dtype=np.uint64)
        return (np.frombuffer(joined, dtype=np.uint8) if joined else np.zeros(1, np.uint8)), off

    def mate1(self, d):
        """reads_1[d] (PE) / reads[d] (SE) of the strand-doubled input (shannon.py:396-424)."""
        if self.r2 is None or d < self.n:
            s = self._get(self.r1, d if d < self.n else d - self.n)
            return s if d < self.n else self._rc(s)
        return self._rc(self._get(self.r2, d - self.n))

    def mate2(self, d):
        if d < self.n:
            return self._rc(self._get(self.r1, d))
        return self._get(self.r2, d - self.n)


_RC = str.maketrans("ACGTN", "TGCAN")
