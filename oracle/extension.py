"""Contig extension / k1-mer error correction / contig-graph components (rows a3-a7).
Test infrastructure (see oracle/__init__.py).  Follows extension_correction.py.

NOTE the reference's `K` inside extension_correction.py is the (K+1)-mer length
(extension_correction.py:220); it is called `k1` here.
"""
import math

BASES = ["A", "G", "C", "T"]          # extension_correction.py:10 (tie order of argmax)


def low_complexity(kmer):
    """extension_correction.py:142-149: most frequent base occurs >= len-2 times."""
    return max(kmer.count(c) for c in "ACGT") >= len(kmer) - 2


def load_kmers(items):
    """extension_correction.py:202-221 (double_stranded=False, polyA_del=True).
    `items`: iterable of (kmer, count) in file order.  Returns ({kmer: float}, k1)."""
    kmers = {}
    for kmer, weight in items:
        kmer = kmer.upper()
        if low_complexity(kmer):
            continue
        kmers[kmer] = kmers.get(kmer, 0) + float(weight)
    k1 = len(next(iter(kmers))) if kmers else 0
    return kmers, k1


def _extend(start, right, traversed, kmers, k1):
    """extension_correction.py:223-245: greedy walk; among untraversed existing neighbours pick
    max weight, ties -> first of A,G,C,T (strict > in argmax, :159-166)."""
    last = start[-(k1 - 1):] if right else start[:k1 - 1]
    ext, tot_w, tot_n = [], 0, 0
    while True:
        best, best_w, best_k = None, None, None
        for b in BASES:
            cand = last + b if right else b + last
            if cand in kmers and cand not in traversed:
                w = kmers[cand]
                if best is None or w > best_w:
                    best, best_w, best_k = b, w, cand
        if best is None:
            return ext, tot_w, tot_n
        ext.append(best)
        tot_w += best_w
        tot_n += 1
        traversed.add(best_k)
        last = best_k[-(k1 - 1):] if right else best_k[:k1 - 1]


class ExtensionResult(object):
    pass


def python_walks(kmers, k1, min_weight):
    """extension_correction.py:334-354: the seed loop.  Yields (contig, total weight, number of k1-mers) per seed that is not
    traversed yet, heaviest first."""
    heaviest = sorted(kmers.items(), key=lambda kv: kv[1])          # :334 (stable)
    traversed = set()
    while heaviest:
        start, w = heaviest.pop()                                   # :344
        if w < min_weight:
            break
        if start in traversed:
            continue
        traversed.add(start)
        rext, rw, rn = _extend(start, True, traversed, kmers, k1)
        lext, lw, ln = _extend(start, False, traversed, kmers, k1)
        yield "".join(reversed(lext)) + start + "".join(rext), rw + lw + kmers[start], rn + ln + 1


def run_correction(items, min_weight=3, min_length=75, comp_size_threshold=500, r=15, f=0.5, walks=None):
    """extension_correction.py:309-524.  walks: the seed loop's products from elsewhere (the C restatement oracle/ext_c.c through
    build_c.extend: the same walks, found in seconds where the Python loop takes minutes) instead of python_walks.
    Returns an ExtensionResult with
      contigs            accepted contigs in acceptance order (== k1mer.dict_contig lines)
      allowed            {k1mer: int weight}                      (:404-408)
      connections        {idx: {idx2: w}} 1-based contig indices  (:372-389)
      components         {root: [idx...]} DFS order               (:417-434)
      single_contigs     [contig]  -> reconstructed_single_contigs.fasta (>Single_i) (:467-473)
      big_components     [(contig list, metis_text)] -> component{n}.txt / component{n}contigs.txt
      remaining          [[contig...], ...] -> remaining_contigs{r}.txt (possibly a trailing [])
    """
    kmers, k1 = load_kmers(items)
    allowed = set()
    rmer_to_contig, cmer_to_contig = {}, {}
    contig_connections = {}
    contigs = ["buffer"]
    contig_index = 0
    for contig, tot_wt, tot_kmer in (walks if walks is not None else python_walks(kmers, k1, min_weight)):
        avg_wt = tot_wt / max(1, tot_kmer)
        # duplicate_check, :247-270
        dup_count = {}
        max_till_now, max_idx = 0, -1
        L = len(contig)
        for i in range(0, L - r + 1):
            lst = rmer_to_contig.get(contig[i:i + r])
            if lst is not None:
                for dup in lst:
                    c = dup_count.get(dup, 0) + 1
                    dup_count[dup] = c
                    if c >= max_till_now:
                        max_till_now, max_idx = c, dup
        covered = [0] * L
        for i in range(0, L - r + 1):
            lst = rmer_to_contig.get(contig[i:i + r])
            if lst is not None and max_idx in lst:
                for j in range(i, i + r):
                    covered[j] = 1
        duplicate_suspect = sum(covered) > f * float(L)
        # hyperbola filter, :361
        if (L >= min_length and L * math.pow(avg_wt, 1 / 4.0) >= 2 * min_length * math.pow(min_weight, 1 / 4.0)
                and not duplicate_suspect):
            contig_index += 1
            contigs.append(contig)
            contig_connections.setdefault(contig_index, {})
            for i in range(L - k1 + 1):
                allowed.add(contig[i:i + k1])
            C = k1 - 1
            for i in range(L - C + 1):
                cm = contig[i:i + C]
                if cm in cmer_to_contig:
                    for c2 in cmer_to_contig[cm]:
                        if c2 != contig_index:
                            d = contig_connections[contig_index]
                            d[c2] = d.get(c2, 0) + 1
                            d2 = contig_connections[c2]
                            d2[contig_index] = d2.get(contig_index, 0) + 1
                else:
                    cmer_to_contig[cm] = []
                cmer_to_contig[cm].append(contig_index)
            for i in range(L - r + 1):
                rmer_to_contig.setdefault(contig[i:i + r], []).append(contig_index)

    res = ExtensionResult()
    res.k1 = k1
    res.contigs = contigs[1:]
    res.allowed = {k: int(kmers[k]) for k in allowed}
    res.connections = contig_connections

    # DFS components, :417-434
    contig2component, component2contig, seen = {}, {}, {}
    for ci in contig_connections:
        if ci not in contig2component:
            component2contig[ci] = []
            stack = [ci]
            seen[ci] = True
            while stack:
                cur = stack.pop()
                contig2component[cur] = ci
                component2contig[ci].append(cur)
                for nb in contig_connections[cur]:
                    if nb not in seen:
                        stack.append(nb)
                        seen[nb] = True
    res.components = component2contig
    connections_drawn = {c: set() for c in component2contig}
    for a in contig_connections:
        for b in contig_connections[a]:
            connections_drawn[contig2component[a]].add((min(a, b), max(a, b)))

    # file emit, :458-513
    res.single_contigs, res.big_components = [], []
    res.remaining = [[]]
    cur_size = 0
    for comp, members in component2contig.items():
        if len(members) == 1:
            res.single_contigs.append(contigs[members[0]])
            continue
        if len(members) > comp_size_threshold:
            code = {c: i + 1 for i, c in enumerate(members)}
            lines = ["%d\t%d\t001\n" % (len(members), len(connections_drawn[comp]))]
            for c in members:
                lines.append("".join("%d\t%d\t" % (code[c2], wt) for c2, wt in contig_connections[c].items()) + "\n")
            res.big_components.append(([contigs[c] for c in members], "".join(lines)))
        else:
            for c in members:
                res.remaining[-1].append(contigs[c])
            cur_size += len(members)
            if cur_size > comp_size_threshold:
                res.remaining.append([])
                cur_size = 0
    return res


def write_outputs(res, directory):
    """Writes the same file tree as extension_correction.py:458-513 into `directory`."""
    import os
    with open(os.path.join(directory, "reconstructed_single_contigs.fasta"), "w") as f:
        for i, c in enumerate(res.single_contigs):
            f.write(">Single_%d\n%s\n" % (i, c))
    for n, (cl, metis) in enumerate(res.big_components):
        open(os.path.join(directory, "component%d.txt" % (n + 1)), "w").write(metis)
        open(os.path.join(directory, "component%dcontigs.txt" % (n + 1)), "w").write("".join(c + "\n" for c in cl))
    for n, cl in enumerate(res.remaining):
        open(os.path.join(directory, "remaining_contigs%d.txt" % (n + 1)), "w").write("".join(c + "\n" for c in cl))
