"""Partition naming, k1mer->partition table, read routing, per-partition k1-mer emit
(rows a8-a11).  Test infrastructure (see oracle/__init__.py).  Follows
kmers_for_component.py and weight_updated_graph.py.  gpmetis itself is external (METIS 5,
unpinned): the partition vectors are inputs here.
"""
import math


def n_partitions(num_contigs, partition_size):
    """kmers_for_component.py:217."""
    return min(int(math.ceil(float(num_contigs) / float(partition_size))), 100)


def weight_updated_graph(metis_text, part, penalty=5):
    """weight_updated_graph.py:24-42: multiply the weight of every edge cut by `part`
    (list of partition ids, one per vertex) by `penalty`.  Returns the new METIS text."""
    lines = metis_text.splitlines()
    out = [lines[0] + "\n"]
    for i, line in enumerate(lines[1:], start=1):
        tok = line.split()
        new = ""
        for j in range(0, len(tok), 2):
            nb = tok[j]
            new += nb + "\t"
            if int(part[i - 1]) != int(part[int(nb) - 1]):
                new += str(penalty * int(tok[j + 1])) + "\t"
            else:
                new += tok[j + 1] + "\t"
        out.append(new + "\n")
    return "".join(out)


def build_partitions(big_components, parts, parts_r2, remaining, allowed, K):
    """kmers_for_component.py:244-305.
      big_components : [contig list] per gpmetis'd component (file order)
      parts/parts_r2 : [partition-id list] per component (first / second gpmetis run; r2 may be None)
      remaining      : [contig list] per remaining_contigs{r}.txt
      allowed        : {k1mer: int}
    Returns (new_components {name: [contig...]} in creation order,
             k1mers2component {k1mer: [set(names), weight]})."""
    k1 = K + 1
    new_components, k2c = {}, {}

    def add(comp, contig):
        new_components.setdefault(comp, []).append(contig)
        for e in range(len(contig) - k1 + 1):
            km = contig[e:e + k1]
            if km not in k2c:
                k2c[km] = [set([comp]), allowed.get(km, 0)]
            else:
                k2c[km][0].add(comp)

    for i, contigs in enumerate(big_components):
        for j, pid in enumerate(parts[i]):
            add("c%d_%s" % (i + 1, pid), contigs[j])
        if parts_r2 is not None:
            for j, pid in enumerate(parts_r2[i]):
                add("r2_c%d_%s" % (i + 1, pid), contigs[j])
    for i, contigs in enumerate(remaining):
        for c in contigs:
            add("cremaining%d" % (i + 1), c)
    return new_components, k2c


def get_rmers(read, R):
    """kmers_for_component.py:186-192: windows at 0,R,2R,... while i < len-R, plus the last."""
    i, out = 0, []
    while i < len(read) - R:
        out.append(read[i:i + R])
        i += R
    out.append(read[-R:])
    return out


def get_comps(read, k2c, K):
    """kmers_for_component.py:194-205: union of partitions hit by the probes."""
    s = set()
    for km in get_rmers(read, K + 1):
        e = k2c.get(km)
        if e is not None:
            s |= e[0]
    return s


def route_reads(reads, new_components, k2c, K):
    """SE routing, kmers_for_component.py:322-355 (double_stranded=False: reads already doubled).
    Reads with any char outside ACTG are dropped (:336).  Returns {comp: [read...]} in input order."""
    out = {c: [] for c in new_components}
    for r in reads:
        if r.strip("ACTG"):
            continue
        for c in get_comps(r, k2c, K):
            out[c].append(r)
    return out


def route_reads_paired(reads1, reads2, new_components, k2c, K):
    """PE routing, kmers_for_component.py:358-403: pair goes to the union over both mates."""
    o1 = {c: [] for c in new_components}
    o2 = {c: [] for c in new_components}
    for a, b in zip(reads1, reads2):
        if a.strip("ACTG") or b.strip("ACTG"):
            continue
        for c in get_comps(a, k2c, K) | get_comps(b, k2c, K):
            o1[c].append(a)
            o2[c].append(b)
    return o1, o2


def partition_k1mers(new_components, k2c, K):
    """kmers_for_component.py:452-477: per partition, per contig, per k1-window in order:
    (k1mer, weight).  Returns {comp: [(k1mer, w), ...]} (== component{comp}k1mers_allowed.dict)
    and contig_weights {comp: [[w...] per contig]} (the --inMem form, :468-469)."""
    k1 = K + 1
    files, cw = {}, {}
    for comp, contigs in new_components.items():
        rows, ws = [], []
        for contig in contigs:
            wl = []
            for i in range(len(contig) - k1 + 1):
                km = contig[i:i + k1]
                w = k2c[km][1]
                wl.append(w)
                rows.append((km, w))
            ws.append(wl)
        files[comp] = rows
        cw[comp] = ws
    return files, cw


def route_pairs_matrix(r1, r2, new_components, K):
    """route_reads_paired over equal-length reads held as code matrices (uint8 [n, L], values 0..3; anything else = a base outside
    ACGT: the pair is dropped, :376), vectorised: the strand-doubled pair d is (R1[d], RC(R2'[d]))... exactly the files of
    shannon.py:413-424 -- mate 1 of pair d < n is R1[d], of pair d >= n it is RC(R2[d - n]); mate 2 is RC(R1[d]) / R2[d - n] -- and
    the probes of a mate are its k1-windows at 0, k1, 2 k1, ... and its last one (get_rmers).  Returns {comp: ascending doubled
    indices of the pairs routed to it} -- the same pairs, in the same (file) order, as route_reads_paired gives as strings
    (tests/test_oracle_c.py)."""
    import numpy as np
    k1 = K + 1
    n, L = r1.shape
    names = list(new_components)
    pw = (np.uint64(4) ** np.arange(k1 - 1, -1, -1, dtype=np.uint64)).astype(np.uint64)
    keys, owner = [], []
    code = np.zeros(256, np.uint64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    for ci, comp in enumerate(names):
        for contig in new_components[comp]:
            if len(contig) >= k1:
                c = code[np.frombuffer(contig.encode(), np.uint8)]
                kk = np.lib.stride_tricks.sliding_window_view(c, k1) @ pw
                keys.append(kk)
                owner.append(np.full(len(kk), ci, np.int64))
    out = {c: np.zeros(0, np.int64) for c in names}
    if not keys or L < k1:
        return out
    keys, owner = np.concatenate(keys), np.concatenate(owner)
    o = np.argsort(keys, kind="stable")
    keys, owner = keys[o], owner[o]
    starts = list(range(0, L - k1, k1)) + [L - k1]                      # get_rmers: while i < len - R, then the last
    # (the strand-doubled pair d < n is R1[d] with its own reverse complement, pair n + i is RC(R2[i]) with R2[i]: a base outside
    # ACGT drops the pairs made of THAT read)
    hits = []                                                            # (doubled index, comp) of every probe that hits
    for mat, fwd_base, rc_base in ((r1, 0, n), (r2, n, 0)):             # a forward probe of R1[i] belongs to pair i (mate 1), a probe of RC(R1[i]) to pair i too (mate 2);
        m64 = np.minimum(mat, 3).astype(np.uint64)                      # of R2[i]: forward -> pair n + i (mate 2), RC -> pair n + i (mate 1)
        good = (mat < 4).all(axis=1)
        for p in starts:
            fw = m64[:, p:p + k1] @ pw
            q = L - k1 - p                                               # RC(read)[p : p + k1] = rc(read[q : q + k1])
            rv = (np.uint64(3) - m64[:, q:q + k1]) @ pw[::-1]
            for kk in (fw, rv):
                pos = np.searchsorted(keys, kk)
                pos2 = np.minimum(pos, len(keys) - 1)
                hit = (keys[pos2] == kk) & good
                idx = np.nonzero(hit)[0]
                # a k1-mer may lie in several partitions (the r2_ round): all the entries with this key
                while len(idx):
                    hits.append(np.stack([idx + (fwd_base if mat is r1 else n), owner[pos2[idx]]], axis=1))
                    pos2 = pos2.copy()
                    nxt = pos2[idx] + 1
                    ok = (nxt < len(keys))
                    ok[ok] &= keys[nxt[ok]] == kk[idx[ok]]
                    idx = idx[ok]
                    pos2[idx] = nxt[ok]
    if not hits:
        return out
    h = np.unique(np.concatenate(hits), axis=0)
    for ci, comp in enumerate(names):
        out[comp] = np.sort(h[h[:, 1] == ci, 0])
    return out
