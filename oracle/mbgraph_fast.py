"""The multibridged-graph oracle (oracle/mbgraph.py) with its two loops over ALL reads vectorised -- TEST INFRASTRUCTURE like the rest
of this package.  The graph, every decision on it and every other loop are oracle/mbgraph.py's own code (this is a subclass); only

  * load_reads / load_mated_reads  (multibridging.py:22-30, 68-97 -> mbgraph.py:44-62: one dict probe per read) and
  * find_bridging_reads            (mbgraph.py:88-111: one dict probe per interior K-mer window of every distinct read)

are restated over uint8 matrices with numpy, producing the SAME Read objects (insertion numbers, copy counts, mates, mate_pair
flags as the sequential loop leaves them -- the last occurrence of a read decides its mate) and the same (read, start) lists on the
X-nodes in the same order.  tests/test_oracle_fast.py holds it against oracle/mbgraph.py itself on whole partitions; with it a
partition of millions of read pairs (the largest of BASELINE configs[2]: 3.9 M pairs, the one that sets the graph stage's wall
time) goes through the oracle in about a minute instead of the pure-Python loops' ~80 us per pair.
Reads must be of one length (a matrix); anything else takes oracle/mbgraph.py."""
import numpy as np
from .mbgraph import MBGraph, Read

_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def _upper(X):
    """str.upper() over ASCII codes"""
    low = (X >= 97) & (X <= 122)
    return np.where(low, X - 32, X).astype(np.uint8) if low.any() else X


def _distinct_rows(X):
    """rows of the uint8 matrix X grouped by equality: (group of every row numbered by first occurrence, index of the first row of
    every group, rows per group).  Grouping by two independent 64-bit hashes, then every row is compared with its group's first."""
    n, L = X.shape
    rng = np.random.Generator(np.random.PCG64(12345))
    m1 = rng.integers(1, 1 << 62, size=L, dtype=np.uint64) | np.uint64(1)
    m2 = rng.integers(1, 1 << 62, size=L, dtype=np.uint64) | np.uint64(1)
    h = np.empty(n, dtype=[("a", np.uint64), ("b", np.uint64)])
    step = 1 << 18
    for s in range(0, n, step):
        blk = X[s:s + step].astype(np.uint64)
        h["a"][s:s + step] = (blk * m1).sum(axis=1, dtype=np.uint64)
        h["b"][s:s + step] = ((blk + np.uint64(7)) * m2).sum(axis=1, dtype=np.uint64)
    _u, first, inv, cnt = np.unique(h, return_index=True, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    for s in range(0, n, step):                                     # exactness: a hash collision would merge different reads
        if not np.array_equal(X[s:s + step], X[first[inv[s:s + step]]]):
            raise RuntimeError("mbgraph_fast: 128-bit row hashes collided; use oracle/mbgraph.py")
    order = np.argsort(first, kind="stable")                        # groups in order of first occurrence = insertion numbers
    rank = np.empty(len(first), dtype=np.int64)
    rank[order] = np.arange(len(first))
    return rank[inv], first[order], cnt[order]


class FastMBGraph(MBGraph):
    def _make_reads(self, X, n_pairs_or_reads, mated):
        """the Read objects the sequential loader leaves behind for the rows X (ASCII, upper case), in insertion order"""
        rid, first, cnt = _distinct_rows(X)
        nd = len(first)
        text = X[first].tobytes().decode("ascii")
        L = X.shape[1]
        reads = [Read(d, text[d * L:(d + 1) * L], float(c)) for d, c in enumerate(cnt.tolist())]
        if mated:
            last = np.zeros(nd, dtype=np.int64)
            np.maximum.at(last, rid, np.arange(len(rid), dtype=np.int64))        # the last occurrence sets mate and mate_pair (:93-96)
            mate = rid[last ^ 1]
            for r, p, m in zip(reads, last.tolist(), mate.tolist()):
                r.mate_pair = 1 if (p & 1) == 0 else 2
                r.mate = reads[m]
        self.reads = {r.bases: r for r in reads}
        self._rows = X[first]                                        # the distinct reads' text, row d = read d
        return reads

    def load_read_rows(self, A):
        """multibridging.py:22-30 over a matrix of ASCII rows"""
        cutoff = len(self.nodes) * 10
        A = np.ascontiguousarray(A[:cutoff + 1])
        self._make_reads(_upper(A), len(A), False)

    def load_mated_rows(self, A, B):
        """multibridging.py:68-97 over two matrices of ASCII rows (mate 1 / mate 2 of every pair)"""
        cutoff = len(self.nodes) * 10
        self.mated = True
        n = min(len(A), len(B), cutoff + 1)
        X = np.empty((2 * n, A.shape[1]), dtype=np.uint8)
        X[0::2] = A[:n]
        X[1::2] = B[:n]
        self._make_reads(_upper(X), n, True)

    def find_bridging_reads(self):
        """mbgraph.py:88-111: every interior K-mer window of every distinct read against the first K-mers of the X-nodes"""
        rows = getattr(self, "_rows", None)
        if rows is None or self.K > 31:
            return MBGraph.find_bridging_reads(self)
        K = self.K
        starts = {}
        for n in self.nodes:
            if n.is_xnode():
                starts.setdefault(n.bases[:K], []).append(n)
        if not starts or not len(rows):
            return
        def key_of(s):
            v = 0
            for ch in s:
                c = int(_CODE[ord(ch)])
                if c > 3:
                    return None
                v = (v << 2) | c
            return v
        by_key = {}
        for s, xs in starts.items():
            k = key_of(s)
            if k is not None:                                       # (a K-mer with a base outside ACGT matches only itself: none of the reads' windows below)
                by_key[k] = xs
        other = {s: xs for s, xs in starts.items() if key_of(s) is None}
        xkeys = np.array(sorted(by_key), dtype=np.uint64)
        reads = list(self.reads.values())
        L = rows.shape[1]
        nw = L - K - 1                                              # starts 1 .. L - K - 1
        if nw <= 0:
            return
        step = 1 << 17
        for s0 in range(0, len(rows), step):
            codes = _CODE[rows[s0:s0 + step]]
            bad = (codes > 3)
            c64 = (codes & 3).astype(np.uint64)
            key = np.zeros((len(codes), nw), dtype=np.uint64)
            badw = np.zeros((len(codes), nw), dtype=bool)
            for j in range(K):
                key = (key << np.uint64(2)) | c64[:, 1 + j:1 + j + nw]
                badw |= bad[:, 1 + j:1 + j + nw]
            pos = np.searchsorted(xkeys, key)
            hit = (xkeys[np.minimum(pos, len(xkeys) - 1)] == key) & ~badw if len(xkeys) else np.zeros_like(badw)
            ri, si = np.nonzero(hit)                                # row-major: reads in insertion order, starts ascending
            for r_i, s_i, k_i in zip(ri.tolist(), si.tolist(), key[ri, si].tolist()):
                read = reads[s0 + r_i]
                for x in by_key[k_i]:
                    if read.bridges(x, s_i + 1):
                        x.reads.append((read, s_i + 1))
            if other:                                               # X-node K-mers with other characters: the sequential rule for them
                for r_i in range(len(codes)):
                    read = reads[s0 + r_i]
                    for start in range(1, len(read.bases) - K):
                        xs = other.get(read.bases[start:start + K])
                        if xs:
                            for x in xs:
                                if read.bridges(x, start):
                                    x.reads.append((read, start))


def run_partition_rows(k1mer_rows, A, B, K):
    """oracle.mbgraph.run_partition for reads given as ASCII matrices (A: mate 1 or the single reads, B: mate 2 or None)"""
    L = A.shape[1] if len(A) else 0
    g = FastMBGraph(K, L)
    g.load_k1mers(k1mer_rows)
    if B is not None:
        g.load_mated_rows(A, B)
    else:
        g.load_read_rows(A)
    g.run(True)
    singles, comps = g.output_components()
    return g, singles, comps
