"""(K+1)-mer counting oracle (row a2) -- the Jellyfish stand-in.  Test infrastructure.

Reference call site: shannon.py:439-441
    jellyfish count -m K+1 -s 20000000 -c 4 -t nJobs <strand-doubled reads...>
    jellyfish dump -c -t -L 1   ->  k1mer.dict_org   (KMER<TAB>count per line)
Jellyfish itself is a third-party C++ program (>=2.0.0, shannon.py:92-97), not vendored and
not installed here; its contract for this call is "exact count of every k1-window consisting
only of ACGT, over every record of the input files" (no -C: strands are doubled explicitly,
shannon.py:427,436).  Restated two ways: a dict counter (small inputs) and a numpy
sort/run-length version on 2-bit packed keys (A=0,C=1,G=2,T=3, big-endian, so numeric order
== lexicographic order).
"""
import collections
import numpy as np

CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    CODE[ord(_c)] = _i
    CODE[ord(_c.lower())] = _i


def count_k1mers_dict(records, k1):
    """Brute force: {k1mer string: count} over all records (already strand-doubled)."""
    cnt = collections.Counter()
    for s in records:
        s = s.upper()
        for i in range(len(s) - k1 + 1):
            w = s[i:i + k1]
            if w.strip("ACGT"):
                continue
            cnt[w] += 1
    return dict(cnt)


def encode(s):
    return CODE[np.frombuffer(s.encode(), dtype=np.uint8)]


def keys_of_record(codes, k1):
    """All valid packed k1-mer keys (uint64) of one record given its base codes (255 = non-ACGT)."""
    n = len(codes) - k1 + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    c = codes.astype(np.uint64)
    key = np.zeros(n, dtype=np.uint64)
    bad = np.zeros(n, dtype=bool)
    for j in range(k1):
        w = c[j:j + n]
        bad |= (codes[j:j + n] == 255)
        key = (key << np.uint64(2)) | (w & np.uint64(3))
    return key[~bad]


def count_k1mers_packed(records, k1):
    """Sort + run-length count.  Returns (sorted unique uint64 keys, uint64 counts)."""
    parts = [keys_of_record(encode(s), k1) for s in records]
    if not parts:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    keys = np.concatenate(parts)
    return np.unique(keys, return_counts=True)


def count_k1mers_matrix(codes, k1):
    """Fast path for equal-length reads: codes is uint8 [n_reads, L] (values 0..3, 255=N).
    Returns (sorted unique keys, counts) over the given records (caller doubles strands)."""
    n, L = codes.shape
    w = L - k1 + 1
    if w <= 0 or n == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    key = np.zeros((n, w), dtype=np.uint64)
    bad = np.zeros((n, w), dtype=bool)
    for j in range(k1):
        col = codes[:, j:j + w]
        bad |= (col == 255)
        key = (key << np.uint64(2)) | (col.astype(np.uint64) & np.uint64(3))
    return np.unique(key[~bad], return_counts=True)


def key_to_str(key, k1):
    key = int(key)
    return "".join("ACGT"[(key >> (2 * (k1 - 1 - i))) & 3] for i in range(k1))


def str_to_key(s):
    v = 0
    for c in s:
        v = (v << 2) | "ACGT".index(c)
    return v


def rc_key(key, k1):
    """Reverse complement of a packed key (python int)."""
    key = int(key)
    out = 0
    for _ in range(k1):
        out = (out << 2) | (3 - (key & 3))
        key >>= 2
    return out


def write_dict_org(path, table, k1=None):
    """`jellyfish dump -c -t` format; written in KMER-descending order (pinned seed order,
    SURVEY 8c).  `table` is {str: int}."""
    with open(path, "w") as f:
        for kmer in sorted(table, reverse=True):
            f.write("%s\t%d\n" % (kmer, table[kmer]))
