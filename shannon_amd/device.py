"""Host-side handles over the C ABI: Context, Reads (2-bit packed, device resident), Table."""
import ctypes as C
import numpy as np
from . import _lib

ENC_ASCII, ENC_CODES = 0, 1
ALPHA = "ACGT"
ALPHA_BYTES = np.frombuffer(b"ACGTN", dtype=np.uint8)        # (code 4 = a base outside ACGT)


class Context(object):
    def __init__(self, device=0, stream=None, own_workspaces=False):
        """own_workspaces: the stage workspaces (counting ... routing, unitigs) in a set of this context's own -- for a second
        pipeline whose stages run beside those of another context of the process (shn_ctx_own_workspaces)"""
        self.h = C.c_void_p()
        _lib.check(_lib.lib().shn_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(self.h)))
        self.device = device
        if own_workspaces:
            _lib.check(_lib.lib().shn_ctx_own_workspaces(self.h))

    def sync(self):
        _lib.check(_lib.lib().shn_ctx_sync(self.h))

    def timer_reset(self):
        _lib.check(_lib.lib().shn_timer_reset(self.h))

    def timers(self):
        """{name: (ms, n_regions)} of HIP-event timed kernel groups since the last reset."""
        out = {}
        for slot in range(64):
            name = _lib.lib().shn_timer_name(slot).decode()
            if not name:
                continue
            ms, n = C.c_double(), C.c_uint64()
            _lib.check(_lib.lib().shn_timer_ms(self.h, slot, C.byref(ms), C.byref(n)))
            if n.value:
                out[name] = (ms.value, n.value)
        return out

    def timer_bytes(self):
        """{name: algorithmic bytes} the launch sites of the timed kernels have declared since the last reset (shn_timer_bytes)"""
        out = {}
        for slot in range(64):
            name = _lib.lib().shn_timer_name(slot).decode()
            if not name:
                continue
            b = C.c_uint64()
            _lib.check(_lib.lib().shn_timer_bytes(self.h, slot, C.byref(b)))
            if b.value:
                out[name] = int(b.value)
        return out

    LP_RULES = {"vertex": 0, "center": 1}

    def set_lp_rule(self, rule):
        """'center' (default): the trial LPs of path_decompose return the interior-point limit (analytic centre of the optimal
        face); 'vertex': the vertex rule of rounds 1-2 (shn_lp_set_rule)."""
        _lib.check(_lib.lib().shn_lp_set_rule(self.h, self.LP_RULES[rule]))

    def lp_stats(self, reset=False):
        """census of the LP calls of this context (shn_lp_stats)"""
        out = (C.c_uint64 * 8)()
        _lib.check(_lib.lib().shn_lp_stats(self.h, out, 1 if reset else 0))
        names = ("lp_calls", "lp_degenerate", "lp_trials", "lp_degenerate_trials", "newton_steps", "not_converged", "too_large_trials", "rule")
        d = {k: int(v) for k, v in zip(names, out)}
        d["rule"] = "center" if d["rule"] else "vertex"
        return d

    def close(self):
        if self.h:
            _lib.lib().shn_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RaggedCodes(object):
    """Reads of different lengths as base codes (0..3, 4 = other) one after the other + offsets: what shn_reads_ingest_ragged
    leaves on the host.  Indexes like a list of code arrays (ReadStore decodes them where text is needed)."""

    def __init__(self, codes, off):
        self.codes, self.off = codes, off

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        return self.codes[int(self.off[i]):int(self.off[i + 1])]

    @property
    def total_bases(self):
        return int(self.off[-1])

    def take(self, idx):
        """(codes of reads idx one after the other, their offsets uint64[len(idx) + 1])"""
        idx = np.asarray(idx, dtype=np.int64)
        starts = self.off[idx].astype(np.int64)
        lens = (self.off[idx + 1] - self.off[idx]).astype(np.int64)
        o = np.zeros(len(idx) + 1, dtype=np.uint64)
        o[1:] = np.cumsum(lens, dtype=np.uint64)
        total = int(o[-1])
        if not total:
            return np.zeros(1, np.uint8), o
        src = np.arange(total, dtype=np.int64) + np.repeat(starts - o[:-1].astype(np.int64), lens)
        return np.ascontiguousarray(self.codes[src]), o


class Reads(object):
    """A set of reads packed 2 bits/base in HBM (+1 bit/base non-ACGT mask)."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    @classmethod
    def from_strings(cls, ctx, reads):
        joined = "".join(reads).encode()
        offs = np.zeros(len(reads) + 1, dtype=np.uint64)
        if reads:
            offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        buf = np.frombuffer(joined, dtype=np.uint8) if joined else np.zeros(1, np.uint8)
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_reads_create(ctx.h, buf.ctypes.data, offs.ctypes.data, len(reads), 0, ENC_ASCII, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_codes(cls, ctx, codes):
        """codes: uint8 [n_reads, L] with values 0..3 (anything else = non-ACGT)."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n, L = codes.shape
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_reads_create(ctx.h, codes.ctypes.data, None, n, L, ENC_CODES, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def ingest(cls, ctx, source, fmt=0, want_codes=True):
        """A read file (path, .gz too; or its text as bytes / a uint8 array) through shn_reads_ingest: (Reads, code matrix or
        None) for reads of one length, (Reads, RaggedCodes or None) for reads of different lengths (shn_reads_ingest_ragged).
        2-line FASTA or 4-line FASTQ; anything else (multi-line FASTA, a malformed record) raises ShannonError("... unsupported
        ...") and the caller reads the file record by record."""
        if isinstance(source, str):
            if source.endswith(".gz"):
                import gzip
                with gzip.open(source, "rb") as f:
                    text = np.frombuffer(f.read(), dtype=np.uint8)
            else:
                import os
                text = np.memmap(source, dtype=np.uint8, mode="r") if os.path.getsize(source) else np.zeros(0, np.uint8)
        else:
            text = np.frombuffer(source, dtype=np.uint8) if isinstance(source, (bytes, bytearray, memoryview)) else np.ascontiguousarray(source, dtype=np.uint8)
        n, L = C.c_uint64(), C.c_uint32()
        ptr = text.ctypes.data if len(text) else None
        # the code matrix is smaller than the text: room for the text's size is enough (pages never written stay untouched)
        buf = np.empty(max(len(text), 1), dtype=np.uint8) if want_codes else None
        h = C.c_void_p()
        try:
            _lib.check(_lib.lib().shn_reads_ingest(ctx.h if ctx is not None else None, ptr, len(text), int(fmt), buf.ctypes.data if want_codes else None,
                                                   len(buf) if want_codes else 0, C.byref(n), C.byref(L), C.byref(h) if ctx is not None else None))
        except _lib.ShannonError as ex:
            if "reads of different lengths" not in str(ex):
                raise
            return cls._ingest_ragged(ctx, text, fmt)
        codes = buf[:n.value * L.value].reshape(n.value, L.value) if want_codes else None
        return (cls(ctx, h) if ctx is not None else None), codes

    @classmethod
    def _ingest_ragged(cls, ctx, text, fmt):
        n, mx, tot = C.c_uint64(), C.c_uint32(), C.c_uint64()
        ptr = text.ctypes.data if len(text) else None
        L = _lib.lib()
        _lib.check(L.shn_reads_ingest_ragged(None, ptr, len(text), int(fmt), None, 0, None, 0, C.byref(n), C.byref(mx), C.byref(tot), None))       # scan: sizes
        codes = np.empty(max(tot.value, 1), dtype=np.uint8)
        off = np.empty(n.value + 1, dtype=np.uint64)
        h = C.c_void_p()
        _lib.check(L.shn_reads_ingest_ragged(ctx.h if ctx is not None else None, ptr, len(text), int(fmt), codes.ctypes.data, len(codes), off.ctypes.data,
                                             len(off), C.byref(n), C.byref(mx), C.byref(tot), C.byref(h) if ctx is not None else None))
        return (cls(ctx, h) if ctx is not None else None), RaggedCodes(codes[:tot.value], off)

    def __len__(self):
        return int(_lib.lib().shn_reads_count(self.h))

    @property
    def n_invalid(self):
        return int(_lib.lib().shn_reads_n_invalid(self.h))

    @property
    def max_len(self):
        return int(_lib.lib().shn_reads_max_len(self.h))

    def close(self):
        if self.h:
            _lib.lib().shn_reads_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Table(object):
    """(key,count) table of distinct k1-mers on the device."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def __len__(self):
        return int(_lib.lib().shn_table_size(self.h))

    @property
    def total(self):
        return int(_lib.lib().shn_table_total(self.h))

    @property
    def k(self):
        return int(_lib.lib().shn_table_k(self.h))

    @property
    def canonical(self):
        return bool(_lib.lib().shn_table_canonical(self.h))

    def download(self):
        n = len(self)
        keys = np.empty(n, dtype=np.uint64)
        cnts = np.empty(n, dtype=np.uint32)
        _lib.check(_lib.lib().shn_table_download(self.ctx.h, self.h, keys.ctypes.data, cnts.ctypes.data))
        return keys, cnts

    def dump(self, lower=1):
        """Content of the reference's k1mer.dict_org (`jellyfish dump -c -t -L lower`,
        shannon.py:441) as sorted (keys, counts) arrays: both strands expanded."""
        n = C.c_uint64(0)
        _lib.check(_lib.lib().shn_table_dump(self.ctx.h, self.h, lower, None, None, C.byref(n)))
        keys = np.empty(n.value, dtype=np.uint64)
        cnts = np.empty(n.value, dtype=np.uint32)
        _lib.check(_lib.lib().shn_table_dump(self.ctx.h, self.h, lower, keys.ctypes.data, cnts.ctypes.data, C.byref(n)))
        return keys[:n.value], cnts[:n.value]

    def filter_lower(self, lower):
        """`jellyfish dump -L lower` as a table (--kmer_hard_cutoff, shannon.py:237-241, 441): a new Table without the k1-mers
        whose count in the reference's input is below `lower`; this one stays as it is (the caller closes it)."""
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_table_filter_lower(self.ctx.h, self.h, int(lower), C.byref(h)))
        return Table(self.ctx, h)

    def lookup(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint32)
        _lib.check(_lib.lib().shn_table_lookup(self.ctx.h, self.h, keys.ctypes.data, len(keys), out.ctypes.data))
        return out

    def device_ptrs(self):
        k, c = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().shn_table_device_ptrs(self.h, C.byref(k), C.byref(c)))
        return k.value, c.value

    def shard(self, n_ranks, dev_keys_ptr, dev_counts_ptr):
        per = np.zeros(n_ranks, dtype=np.uint64)
        _lib.check(_lib.lib().shn_table_shard(self.ctx.h, self.h, n_ranks, per.ctypes.data_as(_lib.u64p),
                                              C.c_void_p(dev_keys_ptr), C.c_void_p(dev_counts_ptr)))
        return per

    def shard_by_minimizer(self, n_ranks, dev_keys_ptr, dev_counts_ptr):
        """the pairs grouped by the rank that owns their minimizer (shn_table_shard_mode 1): the shards of shn_cc_*"""
        per = np.zeros(n_ranks, dtype=np.uint64)
        _lib.check(_lib.lib().shn_table_shard_mode(self.ctx.h, self.h, n_ranks, 1, per.ctypes.data_as(_lib.u64p),
                                                   C.c_void_p(dev_keys_ptr), C.c_void_p(dev_counts_ptr)))
        return per

    @classmethod
    def from_pairs(cls, ctx, dev_keys_ptr, dev_counts_ptr, n, k1, canonical):
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_table_from_pairs(ctx.h, C.c_void_p(dev_keys_ptr), C.c_void_p(dev_counts_ptr), n, k1,
                                                   1 if canonical else 0, C.byref(h)))
        return cls(ctx, h)

    def close(self):
        if self.h:
            _lib.lib().shn_table_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ComponentShards(object):
    """Component labelling on owner shards (include/shannon_hip.h: shn_cc_*): the local components of the shard `table` of rank
    `rank` of `world`.  Pointers are device pointers of the caller (torch tensors in distributed.py)."""

    def __init__(self, ctx, table, world, rank):
        self.ctx, self.table, self.world, self.rank = ctx, table, int(world), int(rank)
        self.h = C.c_void_p()
        _lib.check(_lib.lib().shn_cc_create(ctx.h, table.h, self.world, self.rank, C.byref(self.h)))

    def _u64(self, values):
        a = np.zeros(self.world, dtype=np.uint64)
        a[:] = np.asarray(values, dtype=np.uint64)
        return a

    def query_counts(self):
        per = np.zeros(self.world, dtype=np.uint64)
        _lib.check(_lib.lib().shn_cc_query_counts(self.h, per.ctypes.data_as(_lib.u64p)))
        return per

    def queries(self, dev_keys_ptr, dev_labs_ptr):
        _lib.check(_lib.lib().shn_cc_queries(self.h, C.c_void_p(dev_keys_ptr), C.c_void_p(dev_labs_ptr)))

    def answer(self, dev_keys_ptr, dev_labs_ptr, recv_per_rank, base, dev_edges_ptr):
        rp, b = self._u64(recv_per_rank), self._u64(base)
        n = C.c_uint64(0)
        _lib.check(_lib.lib().shn_cc_answer(self.h, C.c_void_p(dev_keys_ptr), C.c_void_p(dev_labs_ptr), rp.ctypes.data_as(_lib.u64p),
                                            b.ctypes.data_as(_lib.u64p), C.c_void_p(dev_edges_ptr), C.byref(n)))
        return int(n.value)

    @staticmethod
    def solve(ctx, dev_edges_ptr, n_edges, id_limit, dev_nodes_ptr, dev_labels_ptr):
        n = C.c_uint64(0)
        _lib.check(_lib.lib().shn_cc_solve(ctx.h, C.c_void_p(dev_edges_ptr), int(n_edges), int(id_limit), C.c_void_p(dev_nodes_ptr),
                                           C.c_void_p(dev_labels_ptr), C.byref(n)))
        return int(n.value)

    def labels(self, base_me, dev_nodes_ptr, dev_labels_ptr, n_nodes, dev_glabel_ptr):
        _lib.check(_lib.lib().shn_cc_labels(self.h, int(base_me), C.c_void_p(dev_nodes_ptr), C.c_void_p(dev_labels_ptr), int(n_nodes),
                                            C.c_void_p(dev_glabel_ptr)))

    def owners(self, dev_glabel_ptr, dev_big_ptr, dev_big_owner_ptr, n_big, dev_owner_ptr):
        _lib.check(_lib.lib().shn_cc_owners(self.h, C.c_void_p(dev_glabel_ptr), C.c_void_p(dev_big_ptr), C.c_void_p(dev_big_owner_ptr),
                                            int(n_big), C.c_void_p(dev_owner_ptr)))

    def shard(self, dev_owner_ptr, dev_keys_ptr, dev_counts_ptr):
        per = np.zeros(self.world, dtype=np.uint64)
        _lib.check(_lib.lib().shn_cc_shard(self.h, C.c_void_p(dev_owner_ptr), per.ctypes.data_as(_lib.u64p), C.c_void_p(dev_keys_ptr),
                                           C.c_void_p(dev_counts_ptr)))
        return per

    def close(self):
        if self.h:
            _lib.lib().shn_cc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def count_k1mers(ctx, read_sets, k1, both_strands=True):
    """Replaces `jellyfish count -m k1` (shannon.py:439).  read_sets: list of Reads."""
    arr = (C.c_void_p * len(read_sets))(*[r.h for r in read_sets])
    h = C.c_void_p()
    _lib.check(_lib.lib().shn_count_k1mers(ctx.h, arr, len(read_sets), k1, 1 if both_strands else 0, C.byref(h)))
    return Table(ctx, h)


def revcomp_keys(keys, k):
    """reverse complements of packed k-mers (numpy; 2 bits per base, first base in the high bits)"""
    keys = np.asarray(keys, dtype=np.uint64).copy()
    out = np.zeros(len(keys), dtype=np.uint64)
    three, two = np.uint64(3), np.uint64(2)
    for _ in range(k):
        out = (out << two) | (three - (keys & three))
        keys >>= two
    return out


def count_k1mers_strand_specific(ctx, d1, d2, k1):
    """The table of a strand-specific run (-s / --ss / --strand_specific): `jellyfish count` without -C (shannon.py:436-439 after
    :427) over `reads` (single-end) or over reads_1 and RC(reads_2) (paired, :407-411) -- forward k1-mers only; the k1-mers of
    RC(reads_2) are the reverse complements of the forward k1-mers of reads_2.  A plain (non-canonical) table."""
    t1 = count_k1mers(ctx, [d1], k1, both_strands=False)
    if d2 is None:
        return t1
    t2 = count_k1mers(ctx, [d2], k1, both_strands=False)
    try:
        # (on the device: the keys of t2 reverse-complemented, both key lists reduced by key -- a k1-mer of reads_1 that is also one
        # of RC(reads_2) gets the sum)
        h = C.c_void_p()
        _lib.check(_lib.lib().shn_table_merge_rc(ctx.h, t1.h, t2.h, C.byref(h)))
    finally:
        t1.close()
        t2.close()
    return Table(ctx, h)


def key_to_str(key, k):
    key = int(key)
    return "".join(ALPHA[(key >> (2 * (k - 1 - i))) & 3] for i in range(k))


def str_to_key(s):
    v = 0
    for c in s:
        v = (v << 2) | ALPHA.index(c)
    return v
