"""Contig extension / error correction / contig-graph components -- host mirror of the
reference's extension_correction.py (rows a3-a7) over the HIP walk kernel.

Device (csrc/extend.hip): load_kmers + lowComplexity filter, seed ordering, the greedy
bidirectional walks (the reference's dominant Python stage, extension_correction.py:334-354).
Host (here): accept filter (:361), duplicate_check (:247-270), allowed set, contig graph by
shared K-mers (:366-397), DFS components and the METIS / contig / remaining-bin files
(:417-513) -- contig-level bookkeeping in the reference's own order.
"""
import ctypes as C
import math
import os, sys
import numpy as np
from . import _lib, device

UNCLAIMED = 0xFFFFFFFF


class ExtensionResult(object):
    """Same fields as the reference's in-memory/file products (see run_correction).  The contig graph is held as CSR
    (conn_off / conn_nb / conn_w, neighbours in the reference's dict insertion order) and the components as flat arrays;
    `connections` ({contig: {neighbour: weight}}) and `components` ({root: [members]}) are the reference's dicts, built on
    first use (tests, small inputs)."""
    _conn = None
    _comps = None

    @property
    def connections(self):
        if self._conn is None:
            off, nb, w = self.conn_off, self.conn_nb, self.conn_w
            self._conn = {a + 1: dict(zip(nb[off[a]:off[a + 1]], w[off[a]:off[a + 1]])) for a in range(len(off) - 1)}
        return self._conn

    @property
    def components(self):
        if self._comps is None:
            m, o = self.comp_members, self.comp_off
            self._comps = {m[o[j]]: m[o[j]:o[j + 1]] for j in range(len(o) - 1)}
        return self._comps


class Extension(object):
    """Handle over shn_ext (device-resident walk state)."""

    BLOCK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int)

    def __init__(self, ctx, table, min_weight=3, max_iterations=0, shard=None, on_block=None):
        """shard = (world, rank): only the walks of the connected components dealt to `rank` (shn_extend_sharded).
        on_block(view, lo, hi, status): called during the call whenever the walks lo <= rank < hi are final (status 0 / 1 =
        last block / -1 = everything handed over so far is void); `view` is an Extension over the unfinished handle, good
        for stats_range / emit / seed_info inside the call-back only (shn_ext_set_block_callback)."""
        self.ctx, self.table = ctx, table
        self.h = C.c_void_p()
        cb = None
        if on_block is not None:
            def _cb(_user, eptr, lo, hi, status):
                on_block(Extension.view(ctx, eptr), int(lo), int(hi), int(status))
            cb = Extension.BLOCK_CB(_cb)
            _lib.lib().shn_ext_set_block_callback(C.cast(cb, C.c_void_p), None)
        try:
            if shard is None or shard[0] <= 1:
                _lib.check(_lib.lib().shn_extend(ctx.h, table.h, int(min_weight), int(max_iterations), C.byref(self.h)))
            else:
                _lib.check(_lib.lib().shn_extend_sharded(ctx.h, table.h, int(min_weight), int(max_iterations), int(shard[0]), int(shard[1]),
                                                        C.byref(self.h)))
        finally:
            if cb is not None:
                _lib.lib().shn_ext_set_block_callback(None, None)

    @classmethod
    def view(cls, ctx, handle):
        """A non-owning Extension over a shn_ext pointer (inside a block call-back)."""
        v = cls.__new__(cls)
        v.ctx, v.table, v.h, v.borrowed = ctx, None, C.c_void_p(handle), True
        return v

    def stats_range(self, lo, n):
        nr = np.empty(max(n, 1), np.uint32)
        nl = np.empty(max(n, 1), np.uint32)
        tw = np.empty(max(n, 1), np.uint64)
        if n:
            _lib.check(_lib.lib().shn_ext_stats_range(self.ctx.h, self.h, int(lo), int(n), nr.ctypes.data, nl.ctypes.data, tw.ctypes.data))
        return nr[:n], nl[:n], tw[:n]

    def seed_info(self, ranks):
        """(seed k1-mer key, seed weight) of the given walks: (weight desc, key asc) is the global walk order."""
        ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
        keys = np.empty(max(len(ranks), 1), np.uint64)
        w = np.empty(max(len(ranks), 1), np.uint32)
        if len(ranks):
            _lib.check(_lib.lib().shn_ext_seed_info(self.ctx.h, self.h, ranks.ctypes.data, len(ranks), keys.ctypes.data, w.ctypes.data))
        return keys[:len(ranks)], w[:len(ranks)]

    def digests(self):
        """SHN_EXT_DIGEST=1: the checksums shn_extend kept of its arrays, [8 stages][64 chunks] (table keys, counts, bucket offsets,
        weights + flags, adjacency records, seed order, converged claims, walk records: shn_ext_digests); None without the switch"""
        if os.environ.get("SHN_EXT_DIGEST") != "1":
            return None
        dig = np.zeros(512, np.uint64)
        _lib.check(_lib.lib().shn_ext_digests(self.h, dig.ctypes.data))
        return dig.reshape(8, 64)

    @property
    def n_walks(self):
        return int(_lib.lib().shn_ext_n_walks(self.h))

    @property
    def iterations(self):
        return int(_lib.lib().shn_ext_iterations(self.h))

    @property
    def total_steps(self):
        return int(_lib.lib().shn_ext_total_steps(self.h))

    @property
    def wave_steps(self):
        return int(_lib.lib().shn_ext_wave_steps(self.h))

    @property
    def fresh_steps(self):
        return int(_lib.lib().shn_ext_fresh_steps(self.h))

    @property
    def dense_rounds(self):
        return int(_lib.lib().shn_ext_dense_rounds(self.h))

    @property
    def settled_walks(self):
        return int(_lib.lib().shn_ext_settled_walks(self.h))

    def stats(self):
        n = self.n_walks
        nr = np.empty(n, np.uint32)
        nl = np.empty(n, np.uint32)
        tw = np.empty(n, np.uint64)
        _lib.check(_lib.lib().shn_ext_stats(self.ctx.h, self.h, nr.ctypes.data, nl.ctypes.data, tw.ctypes.data))
        return nr, nl, tw

    def live_stats(self, min_steps=0):
        """(rank, nr, nl, tot_weight) of the non-void walks (of at least min_steps steps), in seed order."""
        L = _lib.lib()
        n = C.c_uint64(0)
        ms = int(max(0, min_steps))
        _lib.check(L.shn_ext_live_stats_min(self.ctx.h, self.h, ms, C.byref(n), None, None, None, None))
        m = n.value
        rank = np.empty(max(m, 1), np.uint32); nr = np.empty(max(m, 1), np.uint32)
        nl = np.empty(max(m, 1), np.uint32); tw = np.empty(max(m, 1), np.uint64)
        if m:
            _lib.check(L.shn_ext_live_stats_min(self.ctx.h, self.h, ms, C.byref(n), rank.ctypes.data, nr.ctypes.data, nl.ctypes.data, tw.ctypes.data))
        return rank[:m], nr[:m], nl[:m], tw[:m]

    def accept(self, k1, min_length, min_weight):
        """(ranks uint32, contig lengths int64) of the walks that pass the accept filter (extension_correction.py:361), in seed
        order: decided on the device (shn_ext_accept), the candidates within 1e-9 of the threshold with math.pow here."""
        L = _lib.lib()
        thr = 2 * min_length * math.pow(min_weight, 1 / 4.0)
        n = C.c_uint64(0)
        _lib.check(L.shn_ext_accept(self.ctx.h, self.h, int(min_length), float(thr), C.byref(n), None, None, None, None))
        m = n.value
        rank = np.empty(max(m, 1), np.uint32); steps = np.empty(max(m, 1), np.uint32)
        tw = np.empty(max(m, 1), np.uint64); cls = np.empty(max(m, 1), np.uint8)
        if m:
            _lib.check(L.shn_ext_accept(self.ctx.h, self.h, int(min_length), float(thr), C.byref(n), rank.ctypes.data, steps.ctypes.data, tw.ctypes.data,
                                        cls.ctypes.data))
        rank, steps, tw, cls = rank[:m], steps[:m], tw[:m], cls[:m]
        keep = cls == 1
        for j in np.nonzero(cls == 2)[0].tolist():
            a = float(int(tw[j])) / max(1, int(steps[j]) + 1)
            keep[j] = (int(steps[j]) + k1) * math.pow(a, 1 / 4.0) >= thr
        if not keep.all():
            rank, steps = rank[keep], steps[keep]
        return rank, steps.astype(np.int64) + k1

    def emit_raw(self, ranks, lengths, reuse=False):
        """(ASCII bases uint8[total], offsets uint64[n+1]) of the contigs of the selected walks.  reuse: the bases land in a
        buffer this thread keeps from call to call (valid until its next such call) -- 300 MB of fresh pages per step cost more
        than the copy into them."""
        ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
        offs = np.zeros(len(ranks) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lengths, dtype=np.uint64)
        need = int(offs[-1]) + 1
        if reuse:
            buf = getattr(_EMIT_TLS, "buf", None)
            if buf is None or len(buf) < need:
                buf = _EMIT_TLS.buf = np.empty(need + need // 8, dtype=np.uint8)
        else:
            buf = np.empty(need, dtype=np.uint8)
        _lib.check(_lib.lib().shn_ext_emit(self.ctx.h, self.h, ranks.ctypes.data, len(ranks), offs.ctypes.data, buf.ctypes.data))
        return buf[:int(offs[-1])], offs

    def emit_device(self, ranks, lengths):
        """(DevText, offsets uint64[n+1]): the contigs of the selected walks, their text left on the device (shn_ext_emit_device)."""
        ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
        offs = np.zeros(len(ranks) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lengths, dtype=np.uint64)
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_ext_emit_device(self.ctx.h, self.h, ranks.ctypes.data, len(ranks), offs.ctypes.data, C.byref(h)))
        return DevText(self.ctx, h), offs

    def emit(self, ranks, lengths):
        buf, offs = self.emit_raw(ranks, lengths)
        return split_strings(buf, offs)

    def weights(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint32)
        _lib.check(_lib.lib().shn_ext_weights(self.ctx.h, self.h, keys.ctypes.data, len(keys), out.ctypes.data))
        return out

    def close(self):
        if self.h and not getattr(self, "borrowed", False):
            _lib.lib().shn_ext_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


import threading as _threading
_EMIT_TLS = _threading.local()

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def windows_to_keys(contig, k):
    """All k-windows of an ACGT string as packed uint64 keys (vectorised)."""
    c = _CODE[np.frombuffer(contig.encode(), dtype=np.uint8)].astype(np.uint64)
    n = len(c) - k + 1
    key = np.zeros(max(n, 0), dtype=np.uint64)
    for j in range(k):
        key = (key << np.uint64(2)) | c[j:j + n]
    return key


def windows_to_keys_many(contigs, k):
    """All k-windows of every string, concatenated in order (one vectorised pass over the joined strings), and the
    number of windows of each string."""
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    nwin = np.maximum(lens - k + 1, 0)
    total = int(lens.sum())
    if total < k or not nwin.any():
        return np.zeros(0, dtype=np.uint64), nwin
    if k <= 32 and total >= 4096:
        try:                                         # native pass (shn_string_windows); non-ACGT text takes the general path below
            keys, _rows, nw = _lib.string_windows(contigs, k)
            return keys, nw
        except _lib.ShannonError:
            pass
    c = _CODE[np.frombuffer("".join(contigs).encode(), dtype=np.uint8)].astype(np.uint64)
    n = total - k + 1
    key = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        key = (key << np.uint64(2)) | c[j:j + n]
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])                  # start of each string in the joined text
    wbefore = np.concatenate([[0], np.cumsum(nwin)[:-1]])
    idx = np.repeat(off - wbefore, nwin) + np.arange(int(nwin.sum()), dtype=np.int64)   # windows that stay inside a string
    return key[idx], nwin


def split_strings(buf, offs):
    s = buf.tobytes().decode()
    o = offs.tolist()
    return [s[o[i]:o[i + 1]] for i in range(len(o) - 1)]


class DevText(object):
    """text left on the device (shn_devtext): the candidate contigs between shn_ext_emit_device and shn_contig_stage_device"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def segments(self, offs, idx):
        """(bytes uint8[sum of lengths], offsets uint64[len(idx)+1]) of the pieces idx of the text cut at offs"""
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        lens = (offs[idx + 1] - offs[idx]) if len(idx) else np.zeros(0, np.uint64)
        out_off = np.zeros(len(idx) + 1, dtype=np.uint64)
        out_off[1:] = np.cumsum(lens, dtype=np.uint64)
        out = np.empty(max(1, int(out_off[-1])), dtype=np.uint8)
        if len(idx):
            _lib.check(_lib.lib().shn_devtext_segments(self.ctx.h, self.h, offs.ctypes.data, len(offs), idx.ctypes.data, len(idx), out.ctypes.data))
        return out[:int(out_off[-1])], out_off

    def close(self):
        if self.h:
            _lib.lib().shn_devtext_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def contig_stage_gpu(ctx, buf, offs, k1, r=15, f=0.5):
    """duplicate_check + contig graph of candidates in seed order with the bulk of the work on the GPU (shn_contig_stage).
    buf/offs: ASCII bases (a host array, or a DevText: shn_contig_stage_device) + offsets.  Returns (acc, best, coff, cnb, cw) as
    numpy arrays (see contig_stage)."""
    n = len(offs) - 1
    acc = np.zeros(max(n, 1), dtype=np.int32)
    best = np.zeros(max(n, 1), dtype=np.int32)
    h = C.c_void_p()
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    if isinstance(buf, DevText):
        _lib.check(_lib.lib().shn_contig_stage_device(ctx.h, buf.h, offs.ctypes.data, n, int(k1), int(r), float(f), acc.ctypes.data,
                                                      best.ctypes.data, C.byref(h)))
    else:
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        _lib.check(_lib.lib().shn_contig_stage(ctx.h, buf.ctypes.data, offs.ctypes.data, n, int(k1), int(r), float(f), acc.ctypes.data,
                                               best.ctypes.data, C.byref(h)))
    try:
        n_acc, n_conn = C.c_uint64(0), C.c_uint64(0)
        _lib.check(_lib.lib().shn_cgraph_sizes(h, C.byref(n_acc), C.byref(n_conn)))
        coff = np.zeros(n_acc.value + 1, dtype=np.uint64)
        cnb = np.zeros(max(1, n_conn.value), dtype=np.int32)
        cw = np.zeros(max(1, n_conn.value), dtype=np.int32)
        _lib.check(_lib.lib().shn_cgraph_export(h, coff.ctypes.data, cnb.ctypes.data, cw.ctypes.data))
    finally:
        _lib.lib().shn_cgraph_destroy(h)
    return acc[:n], best[:n], coff, cnb[:n_conn.value], cw[:n_conn.value]


def metis_text(members, n_edges, coff, cnb, cw):
    """the component file of extension_correction.py:458-513 (gpmetis' input): "<contigs>\t<edges>\t001", then a line per contig
    in member order -- "<neighbour's 1-based place in the component>\t<weight>\t" per connection, in the connections' order.
    members: 1-based contig ids; coff / cnb / cw: the contig graph's CSR (row c - 1 = contig c).  (Written as a byte array by
    numpy -- the digits of all numbers in one pass per decimal place: a component of 60,000 contigs -- bench.py --config 2p -- took
    0.08 s of % formatting per connection.)"""
    mm = np.asarray(members, dtype=np.int64)
    sz = len(mm)
    code = np.zeros(int(len(coff)) + 1, dtype=np.int64)
    code[mm] = np.arange(1, sz + 1)
    starts, ends = coff[mm - 1].astype(np.int64), coff[mm].astype(np.int64)
    deg = ends - starts
    tot = int(deg.sum())
    # the rows' entries one after the other
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(deg)[:-1]]), deg) + np.arange(tot, dtype=np.int64) if tot else np.zeros(0, np.int64)
    inter = np.empty(2 * tot, dtype=np.int64)
    inter[0::2] = code[cnb[idx]]
    inter[1::2] = cw[idx]
    head = ("%d\t%d\t001\n" % (sz, n_edges)).encode()
    if tot and int(inter.min()) < 0:
        raise ValueError("metis_text: negative number")
    # the text as bytes, written by arrays: a number's digits + a tab; a newline behind a row's last number (a row without
    # connections is a newline alone)
    nd = np.ones(2 * tot, dtype=np.int64)
    p10 = 10
    while tot and int(inter.max()) >= p10:
        nd += inter >= p10
        p10 *= 10
    tok_end = np.cumsum(nd + 1)                                     # end of every number's "digits + tab", rows' newlines not counted
    rows_before = np.repeat(np.arange(sz, dtype=np.int64), 2 * deg)  # newlines in front of a number = the rows in front of its row
    start = tok_end - (nd + 1) + rows_before + len(head)
    total = len(head) + (int(tok_end[-1]) if tot else 0) + sz
    out = np.full(total, 9, dtype=np.uint8)                          # tabs everywhere first
    out[:len(head)] = np.frombuffer(head, dtype=np.uint8)
    v = inter.copy()
    for d in range(int(nd.max()) if tot else 0):
        live = nd > d
        out[(start + nd - 1 - d)[live]] = (48 + v[live] % 10).astype(np.uint8)
        v //= 10
    # a row's newline sits behind its last token: at len(head) + (bytes of the tokens of rows 0..i) + i
    tok_bytes_upto = np.concatenate([[0], tok_end])[np.cumsum(2 * deg)] if tot else np.zeros(sz, dtype=np.int64)
    out[len(head) + tok_bytes_upto + np.arange(sz, dtype=np.int64)] = 10
    return out.tobytes().decode("ascii")


def contig_components(coff, cnb):
    """The reference's DFS components over the connections CSR (shn_contig_components): (comp_of, members, comp_off, comp_edges)."""
    n = len(coff) - 1
    coff = np.ascontiguousarray(coff, dtype=np.uint64)
    cnb = np.ascontiguousarray(cnb, dtype=np.int32)
    comp_of = np.zeros(max(n, 1), np.int32)
    members = np.zeros(max(n, 1), np.int32)
    comp_off = np.zeros(n + 2, np.uint64)
    comp_edges = np.zeros(max(n, 1), np.uint64)
    nc = C.c_uint64(0)
    _lib.check(_lib.lib().shn_contig_components(n, coff.ctypes.data, cnb.ctypes.data if len(cnb) else None, comp_of.ctypes.data, members.ctypes.data,
                                                comp_off.ctypes.data, comp_edges.ctypes.data, C.byref(nc)))
    return comp_of[:n], members[:n], comp_off[:nc.value + 1], comp_edges[:nc.value]


class ContigGraph(object):
    """duplicate_check + contig graph fed in several calls, candidates in seed order (shn_cgraph)."""

    def __init__(self, k1, r=15, f=0.5):
        self.h = C.c_void_p()
        _lib.check(_lib.lib().shn_cgraph_create(int(k1), int(r), float(f), C.byref(self.h)))

    def add(self, strings):
        """-> acc: 1-based accepted index (over all calls) of every candidate, or 0"""
        if not strings:
            self.best = np.zeros(0, np.int32)
            return np.zeros(0, np.int32)
        joined = "".join(strings).encode()
        offs = np.zeros(len(strings) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in strings], dtype=np.uint64)
        buf = np.frombuffer(joined, dtype=np.uint8)
        acc = np.zeros(len(strings), dtype=np.int32)
        self.best = np.zeros(len(strings), dtype=np.int32)      # hit count of every candidate's `best` contig (0: no hit)
        _lib.check(_lib.lib().shn_cgraph_add(self.h, buf.ctypes.data, offs.ctypes.data, len(strings), acc.ctypes.data, self.best.ctypes.data))
        return acc

    def connections(self):
        n_acc, n_conn = C.c_uint64(0), C.c_uint64(0)
        _lib.check(_lib.lib().shn_cgraph_sizes(self.h, C.byref(n_acc), C.byref(n_conn)))
        coff = np.zeros(n_acc.value + 1, dtype=np.uint64)
        cnb = np.zeros(max(1, n_conn.value), dtype=np.int32)
        cw = np.zeros(max(1, n_conn.value), dtype=np.int32)
        _lib.check(_lib.lib().shn_cgraph_export(self.h, coff.ctypes.data, cnb.ctypes.data, cw.ctypes.data))
        return coff.tolist(), cnb[:n_conn.value].tolist(), cw[:n_conn.value].tolist()

    def close(self):
        if self.h:
            _lib.lib().shn_cgraph_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def accept_filter(live, nr, nl, tw, k1, min_length, min_weight, arrays=False):
    """The accept filter of the extension loop (extension_correction.py:361) over non-void walks: ranks + contig lengths of
    the walks that pass, in seed order."""
    steps = nr.astype(np.int64)
    steps += nl
    length = steps + k1
    idx = np.nonzero(length >= min_length)[0]                  # first clause of the accept filter (:361)
    cand, clen = live[idx], length[idx]
    thr = 2 * min_length * math.pow(min_weight, 1 / 4.0)
    # second clause of :361, len * avg_wt**0.25 >= 2*min_length*min_weight**0.25: vectorised with a guard band (two square roots
    # for the fourth root: a few ulp from pow, and four times as fast over 0.7 M candidates); only candidates within 1e-9 (relative)
    # of the threshold are decided with math.pow like the reference.
    ckm = steps[idx] + 1
    ctw = tw[idx]
    avg = ctw / np.maximum(1, ckm)
    lhs = np.sqrt(np.sqrt(avg))
    lhs *= clen
    sure = lhs >= thr * (1 + 1e-9)
    maybe = (~sure) & (lhs >= thr * (1 - 1e-9))
    for j in np.nonzero(maybe)[0].tolist():
        a = float(int(ctw[j])) / max(1, int(ckm[j]))
        sure[j] = int(clen[j]) * math.pow(a, 1 / 4.0) >= thr
    if arrays:                                                 # (hundreds of thousands of candidates: no Python lists)
        return cand[sure].astype(np.uint32), clen[sure].astype(np.int64)
    return list(zip(cand[sure].tolist(), clen[sure].tolist()))


def contig_stage(strings, k1, r=15, f=0.5, want_best=False):
    """duplicate_check + contig graph over candidate contigs in seed order (:358-397), native host code (csrc/contig_host.hip,
    the shn_cgraph handle).  Returns (acc, coff, cnb, cw): acc[i] = 1-based accepted index of candidate i or 0; neighbours of
    accepted contig a (0-based) are cnb[coff[a]:coff[a+1]] (1-based accepted indices, dict insertion order) with weights cw.
    want_best: a fifth item, the hit count of the `best` contig of every candidate (the sharded stage's guard needs it)."""
    if not strings:
        out = (np.zeros(0, np.int32), [0], [], [])
        return out + (np.zeros(0, np.int32),) if want_best else out
    cg = ContigGraph(k1, r, f)
    try:
        acc = cg.add(strings)
        best = cg.best
        coff, cnb, cw = cg.connections()
    finally:
        cg.close()
    if os.environ.get("SHN_DEBUG"):
        pos = np.nonzero(acc)[0]
        dec = np.histogram(pos, bins=10, range=(0, max(1, len(strings))))[0].tolist()
        sys.stderr.write("[contig_graph] candidates %d (%d bases), accepted %d; accepted per decile of the seed order: %s\n"
                         % (len(strings), sum(len(x) for x in strings), len(pos), dec))
    return (acc, coff, cnb, cw, best) if want_best else (acc, coff, cnb, cw)


def rmer_join(ctx, candidates, foreign_contigs, r=15):
    """(candidate index, window start, foreign index) of every r-mer window of a candidate that occurs in a foreign contig,
    once per occurrence -- exactly the increments duplicate_check would make (sort/join on the GPU, shn_rmer_join)."""
    z = np.zeros(0, np.uint32)
    if not candidates or not foreign_contigs:
        return z, z, z
    a = device.Reads.from_strings(ctx, candidates)
    b = device.Reads.from_strings(ctx, foreign_contigs)
    try:
        L = _lib.lib()
        n = C.c_uint64(0)
        _lib.check(L.shn_rmer_join(ctx.h, a.h, b.h, int(r), C.byref(n), None, None, None))
        if not n.value:
            return z, z, z
        hc = np.empty(n.value, np.uint32); hs = np.empty(n.value, np.uint32); hf = np.empty(n.value, np.uint32)
        _lib.check(L.shn_rmer_join(ctx.h, a.h, b.h, int(r), C.byref(n), hc.ctypes.data, hs.ctypes.data, hf.ctypes.data))
        return hc, hs, hf
    finally:
        a.close()
        b.close()


def foreign_interference(ctx, local, acc, best_count, foreign, r=15, f=0.5):
    """Number of local candidates whose duplicate_check decision could differ once the accepted contigs of the OTHER shards
    are taken into account.  local / foreign: [(seed weight, seed key, contig)]; acc[i] != 0: candidate i was accepted by the
    shard-local pass, best_count[i]: hit count of its `best` contig there.  Only foreign contigs EARLIER in the global walk
    order (weight desc, key asc) can matter.  With M = best_count: a rejected candidate is safe while every foreign contig has
    fewer hits than M (it can never be the last to reach the running maximum, :255-259); an accepted one is safe while every
    foreign contig with >= M hits covers at most f*len of it (whichever becomes `best`, the candidate is still accepted)."""
    strings = [c[2] for c in local]
    hc, hs, hf = rmer_join(ctx, strings, [c[2] for c in foreign], r)
    if not len(hc):
        return 0
    wl = np.array([c[0] for c in local], dtype=np.int64); kl = np.array([c[1] for c in local], dtype=np.uint64)
    wf = np.array([c[0] for c in foreign], dtype=np.int64); kf = np.array([c[1] for c in foreign], dtype=np.uint64)
    earlier = (wf[hf] > wl[hc]) | ((wf[hf] == wl[hc]) & (kf[hf] < kl[hc]))
    hc, hs, hf = hc[earlier].astype(np.int64), hs[earlier].astype(np.int64), hf[earlier].astype(np.int64)
    if not len(hc):
        return 0
    pair = hc * np.int64(len(foreign)) + hf
    order = np.lexsort((hs, pair))
    pair, hs, hc = pair[order], hs[order], hc[order]
    head = np.concatenate([[True], pair[1:] != pair[:-1]])
    idx = np.nonzero(head)[0]
    count = np.diff(np.concatenate([idx, [len(pair)]]))
    step = np.minimum(np.diff(hs, append=hs[-1] + r), r)          # bases a window adds before the next one of the same pair starts
    last = np.concatenate([head[1:], [True]])
    step[last] = r
    cover = np.add.reduceat(step, idx)
    pc = hc[idx]                                                   # candidate of every pair
    M = np.asarray(best_count, dtype=np.int64)[pc]
    accepted = np.asarray(acc)[pc] != 0
    lens = np.array([len(x) for x in strings], dtype=np.float64)[pc]
    bad = np.where(accepted, (count >= M) & (cover > f * lens), count >= M)
    return int(len(np.unique(pc[bad])))


def run_correction(ctx, table, min_weight=3, min_length=75, comp_size_threshold=500, r=15, f=0.5, want_allowed=True, timings=None,
                   shard=None, merge=None, gather=None, table_size=None):
    """extension_correction.run_correction (extension_correction.py:309-524) on a device k1-mer
    table.  Returns an ExtensionResult: contigs, allowed {k1mer: int}, connections, components,
    single_contigs, big_components [(contigs, metis_text)], remaining [[contig...]].
    shard = (world, rank) + merge(local) -> global: the walks are sharded by connected component; `merge` receives this
    rank's candidates [(seed weight, seed key, contig)] and returns the candidates of all ranks (any order).
    gather (instead of merge): an object with .world, .rank, .all_gather(obj) -> [obj per rank], .all_reduce_max(int): the
    contig stages are sharded as well -- every rank decides its own candidates, a GPU r-mer join against the other shards'
    accepted contigs proves that none of them could have changed a duplicate_check decision (foreign_interference), else all
    ranks fall back to the global sequential pass.
    table_size: the number of k1-mers of the JOB when `table` is this rank's share of it already (whole components: shard = None
    with a gather) -- the choice between the contig stages must be the same on every rank."""
    import time as _t
    T = timings if timings is not None else {}
    _t0 = [_t.time()]

    def lap(name):
        now = _t.time()
        T[name] = T.get(name, 0.0) + now - _t0[0]
        _t0[0] = now

    k1 = table.k
    UNCL = 0xFFFFFFFF
    # Pipelined (single GPU, no sharding): the rank blocks of the walks become final one after the other, heaviest seeds
    # first -- the order in which the reference's loop finishes contigs.  The candidates of a final block are filtered and
    # emitted inside the walk call (block call-back) and handed to a host thread that runs duplicate_check +
    # contig_connections on them while the GPU iterates on the later blocks.
    pipe = None
    # Large tables (BASELINE configs[2]: 20,000 genes, several 10^5 candidate contigs): the contig stage runs after the walks with
    # its sorts on the GPU (contig_stage_gpu) instead of beside them on one host thread.  SHN_CONTIG_GPU=1 / 0 forces / forbids it.
    _cg = os.environ.get("SHN_CONTIG_GPU", "")
    big = _cg == "1" or (_cg != "0" and (len(table) if table_size is None else int(table_size)) >= 20_000_000)
    gpu_contigs = merge is None and (gather is None or gather.world <= 1) and big
    # ... and on several ranks: the walks sharded by connected component, the candidates of all shards gathered (0.3 GB at BASELINE
    # configs[2]) and merged into the global seed order on every rank, ONE replicated GPU contig stage over them -- instead of every
    # rank walking the whole table (the shards' own contig stage is the sequential host loop: 25 s at that size)
    gpu_sharded = merge is None and gather is not None and gather.world > 1 and big
    if merge is None and not gpu_contigs and not gpu_sharded and os.environ.get("SHN_EXT_PIPELINE", "1") != "0":
        import threading, queue

        class _Pipe(object):
            pass
        pipe = _Pipe()
        pipe.q, pipe.void, pipe.error, pipe.cb_seconds, pipe.accepted, pipe.cg = queue.Queue(), False, None, 0.0, [], None
        pipe.n_blocks, pipe.busy, pipe.t_start = 0, 0.0, _t.time()
        pipe.strings, pipe.ranks, pipe.acc, pipe.best = [], [], [], []     # every candidate handed over, in seed order

        def _worker():
            try:
                pipe.cg = ContigGraph(k1, r, f)
                while True:
                    item = pipe.q.get()
                    if item is None:
                        break
                    w0 = _t.time()
                    acc = pipe.cg.add(item)
                    pipe.accepted += [item[i] for i in np.nonzero(acc)[0].tolist()]
                    pipe.strings += item
                    pipe.acc.append(acc)
                    pipe.best.append(pipe.cg.best)
                    pipe.busy += _t.time() - w0
                    if os.environ.get("SHN_DEBUG"):
                        sys.stderr.write("[pipeline] contig stage: %d candidates (%d bases) %.1f ms, started %.1f ms after the walks began\n"
                                         % (len(item), sum(len(x) for x in item), (_t.time() - w0) * 1e3, (w0 - pipe.t_start) * 1e3))
            except BaseException as ex:                       # surfaced on the main thread after the walks
                pipe.error = ex
        pipe.thread = threading.Thread(target=_worker, daemon=True)
        pipe.thread.start()

        def _on_block(view, lo, hi, status):
            c0 = _t.time()
            try:
                if status < 0:
                    pipe.void = True
                elif not pipe.void and pipe.error is None and hi > lo:
                    bnr, bnl, btw = view.stats_range(lo, hi - lo)
                    alive = np.nonzero(bnr != UNCL)[0]
                    keep_b = accept_filter((alive + lo).astype(np.uint32), bnr[alive], bnl[alive], btw[alive], k1, min_length, min_weight)
                    if keep_b:
                        pipe.ranks += [x[0] for x in keep_b]
                        pipe.q.put(view.emit([x[0] for x in keep_b], [x[1] for x in keep_b]))
                    pipe.n_blocks += 1
                    if os.environ.get("SHN_DEBUG"):
                        sys.stderr.write("[pipeline] block [%d,%d) final %.1f ms after the walks began: %d live walks, %d candidates, call-back %.1f ms\n"
                                         % (lo, hi, (c0 - pipe.t_start) * 1e3, len(alive), len(keep_b), (_t.time() - c0) * 1e3))
            except _lib.ShannonError:
                # a block that is not at its fixpoint after all (its claims do not match its walks): the audit at the end of
                # the walks reopens it; what was handed over is void and the stage runs after the walks
                pipe.void = True
            except BaseException as ex:                       # (an exception cannot cross the C frames of the walk call)
                pipe.error = ex
            pipe.cb_seconds += _t.time() - c0
    if pipe is not None:
        # the call-back needs the interpreter lock while the worker may hold it between its native calls: hand it over
        # quickly (the default switch interval of 5 ms would stall the walk loop once per block)
        _swi = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)
    try:
        ext = Extension(ctx, table, min_weight, shard=shard, on_block=_on_block if pipe is not None else None)
    finally:
        if pipe is not None:
            pipe.q.put(None)
            sys.setswitchinterval(_swi)
    lap("ext.gpu_walks")
    if pipe is not None:
        T["ext.gpu_walks"] -= pipe.cb_seconds
        T["ext.filter+emit (inside the walks)"] = T.get("ext.filter+emit (inside the walks)", 0.0) + pipe.cb_seconds
        pipe.thread.join()
        if pipe.error is not None:
            raise pipe.error
        if pipe.void:                                  # the fixpoint audit reopened blocks: what was handed over is void
            pipe.cg.close()
            pipe = None
    if pipe is not None:
        keep = None
    else:
        # non-void walks long enough for the accept filter's length clause, in seed order (compacted on the GPU)
        if (gpu_contigs or gpu_sharded) and os.environ.get("SHN_EXT_ACCEPT_DEVICE", "1") != "0":
            keep_r, keep_l = ext.accept(k1, min_length, min_weight)           # (the filter itself on the device)
            keep = None
        else:
            live, nr, nl, tw = ext.live_stats(min_length - k1)
            lap("ext.filter.stats")
            if gpu_contigs or gpu_sharded:
                keep_r, keep_l = accept_filter(live, nr, nl, tw, k1, min_length, min_weight, arrays=True)
                keep = None
            else:
                keep = accept_filter(live, nr, nl, tw, k1, min_length, min_weight)
    lap("ext.filter")
    csr = None                                         # (coff, cnb, cw): connections in dict insertion order, 1-based neighbours
    contigs = ["buffer"]
    contig_raw = None
    conn = None
    sharded_contigs = False
    dev_text = None
    if gpu_contigs and not gpu_sharded and len(keep_r) and os.environ.get("SHN_EXT_EMIT_DEVICE", "1") != "0":
        # one rank: the candidates' text never leaves the device (0.3 GB at BASELINE configs[2], once down and once up before);
        # the accepted contigs are fetched afterwards
        dev_text, offs = ext.emit_device(keep_r, keep_l)
        buf = dev_text
        lap("ext.emit")
    elif gpu_contigs or gpu_sharded:
        buf, offs = ext.emit_raw(keep_r, keep_l, reuse=True) if len(keep_r) else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        lap("ext.emit")
    if gpu_contigs or gpu_sharded:
        if gpu_sharded:
            # every rank's candidates -> the same merged list everywhere, in the order of the reference's seed loop
            # (weight descending, seed k1-mer ascending; a seed lies in exactly one shard, so the keys are distinct)
            skey, sw = ext.seed_info(keep_r)
            # (as tensors where the gather offers it -- the ranks of a real job: 0.3 GB of candidate text at BASELINE configs[2], which
            # used to travel as pickled objects)
            mine_ = (np.asarray(sw, dtype=np.int64), np.asarray(skey, dtype=np.uint64), np.asarray(offs, dtype=np.uint64),
                     np.ascontiguousarray(buf[:int(offs[-1])]))
            parts = gather.all_gather_arrays(mine_) if hasattr(gather, "all_gather_arrays") else gather.all_gather(mine_)
            lap("ext.gathers")
            w_all = np.concatenate([p[0] for p in parts])
            k_all = np.concatenate([p[1] for p in parts])
            base, segs = 0, []
            for p in parts:
                segs.append(p[2][:-1].astype(np.uint64) + np.uint64(base))
                base += int(p[2][-1])
            src_off = np.concatenate(segs + [np.array([base], dtype=np.uint64)])
            order = np.lexsort((k_all, -w_all))
            buf, offs = _lib.gather_segments(np.concatenate([p[3] for p in parts]), src_off, order, threads=_lib.host_cpus())
            lap("ext.merge")
        acc, _best, coff, cnb, cw = contig_stage_gpu(ctx, buf, offs, k1, r, f)
        csr = (coff, cnb, cw)
        ai = np.nonzero(acc)[0]
        if dev_text is not None:
            # the accepted contigs only, one after the other: from here on they are "the candidates", all of them accepted
            buf, offs = dev_text.segments(offs, ai)
            dev_text.close()
            ai = np.arange(len(ai), dtype=np.int64)
        buf = np.ascontiguousarray(buf)
        raw = memoryview(buf)                              # (only the accepted tenth is ever turned into strings, slice by slice)
        n_before = len(contigs)
        contigs += [str(raw[a:b], "ascii") for a, b in zip(offs[ai].tolist(), offs[ai + 1].tolist())]
        del raw
        if n_before == 1:
            # the accepted contigs as bytes (candidate buffer, its offsets, accepted candidate per contig): what lays contig text out
            # for the device again (kmers_for_component) gathers segments instead of joining and encoding 80 MB of strings
            contig_raw = (buf, np.asarray(offs, dtype=np.uint64), ai.astype(np.int64))
        strings = None
    else:
        strings = (ext.emit([x[0] for x in keep], [x[1] for x in keep]) if keep else []) if pipe is None else None
    if gpu_contigs or gpu_sharded:
        pass
    elif gather is not None and gather.world > 1:
        if pipe is not None:                           # the shard's contig stage ran beside its walks
            strings = pipe.strings
            skey, sw = ext.seed_info(pipe.ranks)
            acc = np.concatenate(pipe.acc) if pipe.acc else np.zeros(0, np.int32)
            bestc = np.concatenate(pipe.best) if pipe.best else np.zeros(0, np.int32)
            coff, cnb, cw = pipe.cg.connections()
            pipe.cg.close()
            T["ext.contig_graph (beside the walks)"] = T.get("ext.contig_graph (beside the walks)", 0.0) + pipe.busy
            pipe = None
        else:
            skey, sw = ext.seed_info([x[0] for x in keep])
            acc, coff, cnb, cw, bestc = contig_stage(strings, k1, r, f, want_best=True)
        local = list(zip(sw.tolist(), skey.tolist(), strings))
        lap("ext.emit")
        mine = [local[i] for i in np.nonzero(acc)[0].tolist()]              # accepted here, local order
        lap("ext.contig_graph")
        everybody = gather.all_gather(mine)
        lap("ext.gathers")
        foreign = [c for rk, lst in enumerate(everybody) if rk != gather.rank for c in lst]
        unsafe = foreign_interference(ctx, local, acc, bestc, foreign, r, f)
        lap("ext.guard")
        safe = gather.all_reduce_max(1 if unsafe else 0) == 0
        lap("ext.gathers")
        if safe:
            sharded_contigs = True
            conn = {}
            conns = gather.all_gather((coff, cnb, cw))
            lap("ext.gathers")
            items = sorted(((c[0], c[1], rk, j) for rk, lst in enumerate(everybody) for j, c in enumerate(lst)), key=lambda t: (-t[0], t[1]))
            gid = {(rk, j): g + 1 for g, (_w, _k, rk, j) in enumerate(items)}    # global 1-based accepted index
            for _w, _k, rk, j in items:
                contigs.append(everybody[rk][j][2])
            for _w, _k, rk, j in items:
                o, nb_, w_ = conns[rk]
                conn[gid[(rk, j)]] = {gid[(rk, q - 1)]: ww for q, ww in zip(nb_[o[j]:o[j + 1]], w_[o[j]:o[j + 1]])}
            lap("ext.merge")
        else:
            allc = [c for lst in gather.all_gather(local) for c in lst]
            allc.sort(key=lambda c: (-c[0], c[1]))
            strings = [c[2] for c in allc]
    elif merge is not None:
        # candidates of all shards in the global walk order (weight descending, seed k1-mer ascending; :334-345)
        skey, sw = ext.seed_info([x[0] for x in keep])
        allc = merge(list(zip(sw.tolist(), skey.tolist(), strings)))
        allc.sort(key=lambda c: (-c[0], c[1]))
        strings = [c[2] for c in allc]
        lap("ext.emit")
    else:
        lap("ext.emit")

    # duplicate_check + contig graph, sequential over candidates in seed order (:358-397)
    if gpu_contigs or gpu_sharded:
        pass
    elif pipe is not None:                             # already done, beside the walks
        csr = pipe.cg.connections()
        pipe.cg.close()
        contigs += pipe.accepted
        T["ext.contig_graph (beside the walks)"] = T.get("ext.contig_graph (beside the walks)", 0.0) + pipe.busy
    elif not sharded_contigs:
        acc, coff, cnb, cw = contig_stage(strings, k1, r, f)
        for i in np.nonzero(acc)[0].tolist():
            contigs.append(strings[i])
        csr = (coff, cnb, cw)
    if csr is None:                                    # merged shards: the dicts in global order -> CSR
        coff, cnb, cw = [0], [], []
        for a in range(1, len(contigs)):
            d = conn.get(a, {})
            cnb += list(d.keys())
            cw += list(d.values())
            coff.append(len(cnb))
        csr = (coff, cnb, cw)
    lap("ext.contig_graph")
    res = ExtensionResult()
    res.k1 = k1
    res.iterations = ext.iterations
    res.n_walks = ext.n_walks
    res.total_steps = ext.total_steps
    res.wave_steps = ext.wave_steps
    res.fresh_steps = ext.fresh_steps
    res.dense_rounds = ext.dense_rounds
    res.settled_walks = ext.settled_walks
    res.ext_digests = ext.digests()                     # (SHN_EXT_DIGEST=1: bench.py names the stage at which a step differed)
    res.contigs = contigs[1:]
    res.contig_raw = contig_raw
    # allowed k1-mers with their integer weights (:366-369, :404-408): GPU table lookup
    allowed = {}
    if res.contigs and want_allowed:
        keys, _nw = windows_to_keys_many(res.contigs, k1)
        w = ext.weights(keys).tolist()
        p = 0
        for c in res.contigs:
            for i in range(len(c) - k1 + 1):
                allowed[c[i:i + k1]] = w[p]
                p += 1
    res.allowed = allowed
    ext.close()

    # DFS components (:417-434, native: shn_contig_components) and file products (:458-513)
    coff_a = np.asarray(csr[0], dtype=np.uint64)
    cnb_a = np.asarray(csr[1], dtype=np.int32)
    cw_a = np.asarray(csr[2], dtype=np.int32)
    _comp_of, members, comp_off, comp_edges = contig_components(coff_a, cnb_a)
    res.conn_off, res.conn_nb, res.conn_w = coff_a.tolist(), cnb_a.tolist(), cw_a.tolist()
    res.comp_members, res.comp_off = members.tolist(), comp_off.tolist()
    mem, co = res.comp_members, res.comp_off
    sizes = np.diff(comp_off.astype(np.int64)) if len(comp_off) > 1 else np.zeros(0, np.int64)
    res.single_contigs, res.big_components, res.remaining = [], [], [[]]
    res.single_ids = []                               # index in res.contigs of every single contig
    cur_size = 0
    for j, sz in enumerate(sizes.tolist()):
        if sz == 1:
            res.single_contigs.append(contigs[mem[co[j]]])
            res.single_ids.append(mem[co[j]] - 1)
        elif sz > comp_size_threshold:
            mm = mem[co[j]:co[j + 1]]
            res.big_components.append(([contigs[c] for c in mm], metis_text(mm, int(comp_edges[j]), coff_a, cnb_a, cw_a)))
        else:
            res.remaining[-1].extend(contigs[c] for c in mem[co[j]:co[j + 1]])
            cur_size += sz
            if cur_size > comp_size_threshold:
                res.remaining.append([])
                cur_size = 0
    lap("ext.components")
    return res


def write_outputs(res, directory, outfile=None):
    """File tree of extension_correction.py:337,362,458-513."""
    if outfile:
        with open(outfile + "_contig", "w") as f:
            f.write("".join(c + "\n" for c in res.contigs))
    with open(os.path.join(directory, "reconstructed_single_contigs.fasta"), "w") as f:
        for i, c in enumerate(res.single_contigs):
            f.write(">Single_%d\n%s\n" % (i, c))
    for n, (cl, metis) in enumerate(res.big_components):
        open(os.path.join(directory, "component%d.txt" % (n + 1)), "w").write(metis)
        open(os.path.join(directory, "component%dcontigs.txt" % (n + 1)), "w").write("".join(c + "\n" for c in cl))
    for n, cl in enumerate(res.remaining):
        open(os.path.join(directory, "remaining_contigs%d.txt" % (n + 1)), "w").write("".join(c + "\n" for c in cl))
