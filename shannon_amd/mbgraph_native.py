"""ctypes front of the native multibridged-graph stage (csrc/mbgraph_host.hip, shn_mbgraph_run):
same inputs / outputs as shannon_amd.mbgraph.run_partition, which remains the readable Python
specification of the algorithm and is cross-checked against this in tests/test_host_graph.py."""
import ctypes as C
import numpy as np
from . import _lib


def _pack_reads(reads):
    joined = "".join(reads).encode()
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    return (np.frombuffer(joined, dtype=np.uint8) if joined else np.zeros(1, np.uint8)), off


class Unitigs(object):
    """The K-mer graphs of all partitions contracted to unitigs in one batch on the GPU (shn_unitigs_build): replaces
    load_single_jellyfish + the first condense_all of every partition (multibridging.py:145-172, mbgraph.py:479-498)."""

    def __init__(self, ctx, partition_contigs, K, flat_text=None):
        """partition_contigs: [[contig, ...] per partition] (the order of the partition's k1-mer file).  flat_text (optional):
        (text, off, part_of, n_contigs) of exactly these contigs in this order, as kmers_for_component already laid them out for
        the probe table (joining and encoding 80 MB of Python strings a second time was a third of this constructor)"""
        n_flat = sum(len(cl) for cl in partition_contigs)
        if flat_text is not None and flat_text[3] == n_flat and n_flat:
            text, off, part_of = flat_text[0], flat_text[1], flat_text[2]
            flat = range(n_flat)
        else:
            flat = [c for cl in partition_contigs for c in cl]
            text = np.frombuffer("".join(flat).encode(), dtype=np.uint8) if flat else np.zeros(1, np.uint8)
            off = np.zeros(len(flat) + 1, dtype=np.uint64)
            if flat:
                off[1:] = np.cumsum([len(c) for c in flat], dtype=np.uint64)
            part_of = np.repeat(np.arange(len(partition_contigs), dtype=np.uint32), [len(cl) for cl in partition_contigs]) if flat else np.zeros(1, np.uint32)
            part_of = np.ascontiguousarray(part_of, dtype=np.uint32)
        self.h = C.c_void_p()
        self.n_parts = len(partition_contigs)
        _lib.check(_lib.lib().shn_unitigs_build(ctx.h, text.ctypes.data, off.ctypes.data, len(flat), part_of.ctypes.data, self.n_parts, int(K),
                                                C.byref(self.h)))

    def n_kmers(self, part):
        return int(_lib.lib().shn_unitigs_n_kmers(self.h, int(part)))

    def close(self):
        if self.h:
            _lib.lib().shn_unitigs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphHandle(object):
    """The multibridged graph of one partition as the native object (shn_graph): exported to Python tables on demand,
    consumed in place by the native sparse-flow stage (sparse_flow_native)."""

    def __init__(self, h):
        self.h = h
        self._tables = None

    def tables(self):
        """(singles, comps, log) in the format of mbgraph.output_components"""
        if self._tables is None:
            self._tables = _export(self.h)
        return self._tables

    def close(self):
        if self.h:
            _lib.lib().shn_graph_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _export(h):
    L = _lib.lib()
    sz = np.zeros(9, dtype=np.uint64)
    _lib.check(L.shn_graph_sizes(h, sz.ctypes.data_as(_lib.u64p)))
    ns, sb, nc, nn, nb, ne, npth, npid, ninfo = [int(x) for x in sz]
    u64, f64, i32, u8 = np.uint64, np.float64, np.int32, np.uint8
    s_off = np.zeros(ns + 1, u64); s_bases = np.zeros(max(sb, 1), u8); s_cc = np.zeros(max(ns, 1), f64); s_norm = np.zeros(max(ns, 1), f64)
    cno = np.zeros(nc + 1, u64); ceo = np.zeros(nc + 1, u64); cpo = np.zeros(nc + 1, u64)
    n_off = np.zeros(nn + 1, u64); n_bases = np.zeros(max(nb, 1), u8); n_cc = np.zeros(max(nn, 1), f64)
    n_cci = np.zeros(max(nn, 1), u8); n_norm = np.zeros(max(nn, 1), f64)
    e_in = np.zeros(max(ne, 1), i32); e_out = np.zeros(max(ne, 1), i32); e_w = np.zeros(max(ne, 1), i32)
    e_cc = np.zeros(max(ne, 1), f64); e_norm = np.zeros(max(ne, 1), f64)
    p_off = np.zeros(npth + 1, u64); p_ids = np.zeros(max(npid, 1), i32); info = np.zeros(max(ninfo, 1), i32)
    arrs = [s_off, s_bases, s_cc, s_norm, cno, ceo, cpo, n_off, n_bases, n_cc, n_cci, n_norm, e_in, e_out, e_w, e_cc, e_norm,
            p_off, p_ids, info]
    _lib.check(L.shn_graph_export(h, *[a.ctypes.data for a in arrs]))
    sbs = s_bases.tobytes().decode()
    singles = [(-1, sbs[int(s_off[i]):int(s_off[i + 1])], 0 if s_cc[i] == 0 else float(s_cc[i]), int(s_norm[i])) for i in range(ns)]
    nbs = n_bases.tobytes().decode()
    noff = n_off.tolist()
    comps = []
    for c in range(nc):
        a, b = int(cno[c]), int(cno[c + 1])
        nodes = [(i - a, nbs[noff[i]:noff[i + 1]], (0 if n_cci[i] else float(n_cc[i])), int(n_norm[i])) for i in range(a, b)]
        ea, eb = int(ceo[c]), int(ceo[c + 1])
        edges = [(int(e_in[i]), int(e_out[i]), int(e_w[i]), float(e_cc[i]), int(e_norm[i])) for i in range(ea, eb)]
        pa, pb = int(cpo[c]), int(cpo[c + 1])
        paths = [p_ids[int(p_off[i]):int(p_off[i + 1])].tolist() for i in range(pa, pb)]
        comps.append({"nodes": nodes, "edges": edges, "paths": paths})
    inf = info.tolist()
    log = ({"nodes_after": inf[:4], "final_nodes": inf[4], "known_paths": inf[5], "mate_paths": inf[6], "bridged": inf[8:8 + inf[7]]}
           if ninfo >= 8 else {})
    return singles, comps, log


def graph_from_tables(singles, comps):
    """GraphHandle from Python tables (the inverse of GraphHandle.tables): [(id, bases, cc, norm)] single rows and
    [{"nodes": [(id, bases, cc, norm)], "edges": [(in, out, w, cc, norm)], "paths": [[ids]]}] components
    (shn_graph_from_tables) -- e.g. the reference's own nodes / edges / paths files."""
    u64, f64, i32, u8 = np.uint64, np.float64, np.int32, np.uint8
    sb = "".join(r[1] for r in singles).encode()
    s_off = np.zeros(len(singles) + 1, u64)
    s_off[1:] = np.cumsum([len(r[1]) for r in singles], dtype=u64) if singles else []
    s_cc = np.array([float(r[2]) for r in singles] + [0.0], f64)
    s_norm = np.array([float(r[3]) for r in singles] + [0.0], f64)
    cno, ceo, cpo, n_len, nb, n_cc, n_cci, n_norm = [0], [0], [0], [], [], [], [], []
    e_in, e_out, e_w, e_cc, e_norm, p_off, p_ids = [], [], [], [], [], [0], []
    for comp in comps:
        for _id, bases, cc, norm in comp["nodes"]:
            nb.append(bases); n_len.append(len(bases)); n_cc.append(float(cc)); n_norm.append(float(norm))
            n_cci.append(1 if (isinstance(cc, int) and cc == 0) else 0)
        for a, b, w, cc, norm in comp["edges"]:
            e_in.append(int(a)); e_out.append(int(b)); e_w.append(int(w)); e_cc.append(float(cc)); e_norm.append(float(norm))
        for pth in comp["paths"]:
            p_ids += [int(x) for x in pth]
            p_off.append(len(p_ids))
        cno.append(len(n_len)); ceo.append(len(e_in)); cpo.append(len(p_off) - 1)
    nbb = "".join(nb).encode()
    n_off = np.zeros(len(n_len) + 1, u64)
    if n_len:
        n_off[1:] = np.cumsum(n_len, dtype=u64)
    sizes = np.array([len(singles), len(sb), len(comps), len(n_len), len(nbb), len(e_in), len(p_off) - 1, len(p_ids), 0], u64)
    arrs = [sizes, s_off, np.frombuffer(sb + b"\0", u8), s_cc, s_norm, np.array(cno, u64), np.array(ceo, u64), np.array(cpo, u64), n_off,
            np.frombuffer(nbb + b"\0", u8), np.array(n_cc + [0.0], f64), np.array(n_cci + [0], u8), np.array(n_norm + [0.0], f64),
            np.array(e_in + [0], i32), np.array(e_out + [0], i32), np.array(e_w + [0], i32), np.array(e_cc + [0.0], f64),
            np.array(e_norm + [0.0], f64), np.array(p_off, u64), np.array(p_ids + [0], i32)]
    h = C.c_void_p()
    _lib.check(_lib.lib().shn_graph_from_tables(*[a.ctypes.data for a in arrs], C.byref(h)))
    return GraphHandle(h)


def sparse_flow_native(ctx, graphs, snames, seed, raw=False, threaded=False):
    """The native sparse-flow stage (shn_sparse_flow) over the graphs of several partitions: [reconstructed FASTA text]
    (raw: as uint8 arrays, undecoded).  threaded: one of several calls running at once, on the calling thread's own stream
    (shn_sparse_flow_thread)."""
    if not graphs:
        return []
    arr = (C.c_void_p * len(graphs))(*[g.h for g in graphs])
    names = (C.c_char_p * len(graphs))(*[s.encode() for s in snames])
    h = C.c_void_p()
    fn = _lib.lib().shn_sparse_flow_thread if threaded else _lib.lib().shn_sparse_flow
    _lib.check(fn(ctx.h, arr, len(graphs), names, C.c_uint64(int(seed)), C.byref(h)))
    try:
        out = []
        for i in range(len(graphs)):
            n = int(_lib.lib().shn_sflow_text_size(h, i))
            buf = np.empty(max(n, 1), np.uint8)
            if n:
                _lib.check(_lib.lib().shn_sflow_text(h, i, buf.ctypes.data))
            out.append(buf[:n] if raw else buf[:n].tobytes().decode())
        return out
    finally:
        _lib.lib().shn_sflow_destroy(h)


def run_partition_rows(ctx, unitigs, part, d1, d2, host1, host2, didx, rows_bytes=None, n_rows=0, routes=None):
    """multibridging.main for partition `part` with its reads named by their rows (shn_mbgraph_run_rows): d1 / d2 the resident
    packed read sets, host1 / host2 the same reads as C-contiguous uint8 code matrices, didx the doubled read indices.
    routes = (Routes still on the device, index of didx[0] in them): the duplicate search reads the list where it lies
    (shn_mbgraph_run_routes)."""
    h = C.c_void_p()
    paired = d2 is not None
    for m in (host1, host2) if paired else (host1,):
        if not (isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.ndim == 2 and m.flags["C_CONTIGUOUS"]):
            raise ValueError("run_partition_rows: host reads must be C-contiguous uint8 code matrices")
    if routes is not None and routes[0].h:
        # didx may be an int: that many entries of the routes from routes[1] on, known by their place on the device only
        n_idx = int(didx) if isinstance(didx, (int, np.integer)) else len(didx)
        hd = None if isinstance(didx, (int, np.integer)) else np.ascontiguousarray(didx, dtype=np.uint32)
        _lib.check(_lib.lib().shn_mbgraph_run_routes(ctx.h, unitigs.h, int(part), rows_bytes.ctypes.data if rows_bytes is not None else None,
                                                     n_rows if rows_bytes is not None else 0, d1.h, d2.h if paired else None, host1.ctypes.data,
                                                     host2.ctypes.data if paired else None, hd.ctypes.data if hd is not None else None, routes[0].h,
                                                     int(routes[1]), n_idx, 1 if paired else 0, C.byref(h)))
        return GraphHandle(h)
    didx = np.ascontiguousarray(didx, dtype=np.uint32)
    _lib.check(_lib.lib().shn_mbgraph_run_rows(ctx.h, unitigs.h, int(part), rows_bytes.ctypes.data if rows_bytes is not None else None,
                                               n_rows if rows_bytes is not None else 0, d1.h, d2.h if paired else None, host1.ctypes.data,
                                               host2.ctypes.data if paired else None, didx.ctypes.data, len(didx), 1 if paired else 0, C.byref(h)))
    return GraphHandle(h)


def reads_dedup(ctx, d1, d2, didx):
    """shn_reads_dedup: (first slot, copies, mate id, role) of the distinct reads among the partition's read slots."""
    didx = np.ascontiguousarray(didx, dtype=np.uint32)
    nm = 2 if d2 is not None else 1
    nh = max(len(didx) * nm, 1)
    slot, cnt, mate, role = np.empty(nh, np.uint32), np.empty(nh, np.uint32), np.empty(nh, np.int32), np.empty(nh, np.uint8)
    nd = C.c_uint64()
    _lib.check(_lib.lib().shn_reads_dedup(ctx.h, d1.h, d2.h if d2 is not None else None, didx.ctypes.data, len(didx), 1 if d2 is not None else 0,
                                          C.byref(nd), slot.ctypes.data, cnt.ctypes.data, mate.ctypes.data, role.ctypes.data))
    n = nd.value
    return slot[:n], cnt[:n], mate[:n], role[:n]


def run_partition_handle(rows_bytes, n_rows, K, r1_buf, r1_off, r2_buf=None, r2_off=None, ctx=None, enc=0, rc1=None, rc2=None, unitigs=None,
                         part=0, resident=None):
    """rows_bytes: uint8 array of n_rows*(K+1) bases; reads as (byte buffer, offsets).  unitigs / part: the partition's
    K-mer graph already contracted on the GPU (rows_bytes may then be None).  resident = (reads1, reads2 or None, doubled read
    indices uint32): the reads are rows of these device-resident sets (shn_mbgraph_run_resident).  Returns the GraphHandle."""
    L = _lib.lib()
    h = C.c_void_p()
    n_reads = len(r1_off) - 1
    paired = r2_buf is not None
    rows_ptr = rows_bytes.ctypes.data if rows_bytes is not None else None
    tail = (r1_buf.ctypes.data, r1_off.ctypes.data, r2_buf.ctypes.data if paired else None, r2_off.ctypes.data if paired else None, n_reads,
            1 if paired else 0, enc, rc1.ctypes.data if rc1 is not None else None, rc2.ctypes.data if rc2 is not None else None, C.byref(h))
    if unitigs is not None and resident is not None and ctx is not None:
        d1, d2, didx = resident
        didx = np.ascontiguousarray(didx, dtype=np.uint32)
        assert len(didx) == n_reads
        _lib.check(L.shn_mbgraph_run_resident(ctx.h, unitigs.h, int(part), rows_ptr, n_rows if rows_bytes is not None else 0, d1.h,
                                              d2.h if d2 is not None else None, didx.ctypes.data, *tail))
    elif unitigs is not None:
        _lib.check(L.shn_mbgraph_run_unitigs(ctx.h if ctx is not None else None, unitigs.h, int(part), rows_ptr, n_rows if rows_bytes is not None else 0, *tail))
    else:
        _lib.check(L.shn_mbgraph_run(ctx.h if ctx is not None else None, K, rows_ptr, n_rows, *tail))
    return GraphHandle(h)


def run_partition_arrays(*args, **kw):
    """run_partition_handle + tables: (singles, comps, info) in the format of mbgraph.output_components."""
    g = run_partition_handle(*args, **kw)
    try:
        return g.tables()
    finally:
        g.close()


def run_partition(k1mer_rows, reads, K, paired=False, ctx=None):
    """Drop-in for mbgraph.run_partition: k1mer_rows = [(k1mer, weight)], reads = [list] or [list1, list2]."""
    rows = np.frombuffer("".join(k for k, _ in k1mer_rows).encode(), dtype=np.uint8) if k1mer_rows else np.zeros(1, np.uint8)
    b1, o1 = _pack_reads(reads[0])
    if paired:
        b2, o2 = _pack_reads(reads[1])
        return run_partition_arrays(rows, len(k1mer_rows), K, b1, o1, b2, o2, ctx=ctx)
    return run_partition_arrays(rows, len(k1mer_rows), K, b1, o1, ctx=ctx)
