"""GPU parity: K-mer seed scans (rows a17, a21) vs the host scan of shannon_amd/mbgraph.py."""
import numpy as np
import pytest
from golden_util import *

pytestmark = pytest.mark.gpu


def test_seed_scans_match_host():
    import ctypes as C
    from shannon_amd import device, _lib, graph_seeds
    from shannon_amd.kmers_for_component import make_table
    ctx = device.Context(0)
    rng = np.random.default_rng(3)
    K = 25
    reads = ["".join("ACGT"[c] for c in rng.integers(0, 4, int(rng.integers(20, 140)))) for _ in range(3000)]
    pats = {}
    for r in reads[::7]:
        if len(r) >= K + 5:
            p = int(rng.integers(0, len(r) - K))
            pats.setdefault(r[p:p + K], len(pats))
    keys = list(pats)
    tab = make_table(ctx, graph_seeds.pack_strings(keys, K), np.arange(1, len(keys) + 1, dtype=np.uint32), K)
    rd = device.Reads.from_strings(ctx, reads)
    n = C.c_uint64(0)
    _lib.check(_lib.lib().shn_seed_scan(ctx.h, rd.h, K, tab.h, C.byref(n), None, None, None))
    r_ = np.empty(n.value, np.uint32); s_ = np.empty(n.value, np.uint32); i_ = np.empty(n.value, np.uint32)
    _lib.check(_lib.lib().shn_seed_scan(ctx.h, rd.h, K, tab.h, C.byref(n), r_.ctypes.data, s_.ctypes.data, i_.ctypes.data))
    ref = [(ri, s, pats[rb[s:s + K]]) for ri, rb in enumerate(reads) for s in range(1, len(rb) - K) if rb[s:s + K] in pats]
    assert list(zip(r_.tolist(), s_.tolist(), i_.tolist())) == ref and len(ref) > 100
    a = np.zeros(len(reads), np.uint32); b = np.zeros(len(reads), np.uint32)
    _lib.check(_lib.lib().shn_seed_ends(ctx.h, rd.h, K, tab.h, a.ctypes.data, b.ctypes.data))
    assert a.tolist() == [pats.get(rb[:K], -1) + 1 if len(rb) >= K else 0 for rb in reads]
    assert b.tolist() == [pats.get(rb[-K:], -1) + 1 if len(rb) >= K else 0 for rb in reads]
    rd.close(); tab.close()
    ctx.close()


@pytest.mark.parametrize("K,seed", [(25, 1), (31, 2), (12, 3)])
def test_known_paths_scan_equals_the_index_walk(K, seed):
    """shn_known_paths_scan against the host rule it replaces (mbgraph.py:1355-1388 / search_sequence :114-160, the part every read
    goes through): look the read's first K-mer up in the index of all K-mers of all nodes (occurrences in node order, then offset);
    an occurrence whose node text equals the read's from there on counts -- inside the node if the read ends before the node does
    (the last such node wins), otherwise the read has to be searched; reads whose first or last K-mer is in no node are skipped."""
    import ctypes as C
    from shannon_amd import device, _lib
    rng = np.random.default_rng(seed)
    ctx = device.Context(0)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    nodes = []
    for _ in range(60):                                              # overlapping pieces of one sequence + repeats of a few
        a = int(rng.integers(0, len(base) - 200)); nodes.append(base[a:a + int(rng.integers(K, 200))])
    nodes += nodes[:5] + [nodes[7][3:], nodes[9][:-4]]
    L = 70
    reads = []
    for _ in range(4000):
        kind = rng.integers(0, 4)
        if kind == 0:                                                # inside a node
            n = nodes[int(rng.integers(0, len(nodes)))]
            if len(n) >= L:
                a = int(rng.integers(0, len(n) - L + 1)); reads.append(n[a:a + L]); continue
        if kind == 1:                                                # from the underlying sequence: often runs past a node's end
            a = int(rng.integers(0, len(base) - L)); reads.append(base[a:a + L]); continue
        if kind == 2:                                                # one substitution
            a = int(rng.integers(0, len(base) - L)); r = list(base[a:a + L]); j = int(rng.integers(0, L)); r[j] = "ACGT"[("ACGT".index(r[j]) + 1) % 4]
            reads.append("".join(r)); continue
        reads.append("".join("ACGT"[i] for i in rng.integers(0, 4, L)))
    index = {}
    for ni, n in enumerate(nodes):
        for o in range(len(n) - K + 1):
            index.setdefault(n[o:o + K], []).append((ni, o))
    want_state, want_node = [], []
    for r in reads:
        st, fn = 0, -1
        if r[:K] in index and r[-K:] in index:
            anyin, need = False, False
            for ni, o in index[r[:K]]:
                n = nodes[ni]
                m = min(len(r), len(n) - o)
                if r[:m] != n[o:o + m]:
                    continue
                if len(r) <= len(n) - o:
                    fn, anyin = ni, True
                else:
                    need = True
                    break
            st = 2 if need else 1 if anyin else 0
        want_state.append(st); want_node.append(fn)
    d = device.Reads.from_strings(ctx, reads)
    text = np.frombuffer("".join(nodes).encode(), np.uint8)
    off = np.zeros(len(nodes) + 1, np.uint64); off[1:] = np.cumsum([len(n) for n in nodes])
    state = np.empty(len(reads), np.uint8); node = np.empty(len(reads), np.int32)
    _lib.check(_lib.lib().shn_known_paths_scan(ctx.h, d.h, K, text.ctypes.data, off.ctypes.data, len(nodes), state.ctypes.data, node.ctypes.data, None))
    assert state.tolist() == want_state
    # with offsets: a read to search whose first K-mer occurs exactly once comes back as 3 with that occurrence
    state3 = np.empty(len(reads), np.uint8); node3 = np.empty(len(reads), np.int32); ofs3 = np.empty(len(reads), np.uint32)
    _lib.check(_lib.lib().shn_known_paths_scan(ctx.h, d.h, K, text.ctypes.data, off.ctypes.data, len(nodes), state3.ctypes.data, node3.ctypes.data,
                                               ofs3.ctypes.data))
    for r, s2, s3, n3, o3 in zip(reads, state.tolist(), state3.tolist(), node3.tolist(), ofs3.tolist()):
        if s2 == 2 and len(index[r[:K]]) == 1:
            assert s3 == 3 and (n3, o3) == index[r[:K]][0]
        else:
            assert s3 == s2
    assert 3 in state3 and 2 in state3
    assert [n if s == 1 else -1 for n, s in zip(node.tolist(), state.tolist())] == [n if s == 1 else -1 for n, s in zip(want_node, want_state)]
    assert set(want_state) == {0, 1, 2}
    d.close(); ctx.close()


@pytest.mark.parametrize("K,seed,cap", [(25, 5, 1 << 16), (31, 6, 1 << 16), (12, 7, 1 << 16), (25, 8, 300)])
def test_known_paths_search_equals_the_recursion(K, seed, cap):
    """shn_known_paths_search: the reads that run past the node their first K-mer lies in, searched on the device, against the
    reference's recursion (mbgraph.py:114-160 search_sequence: follow the out-edges in list order while the node texts agree with
    the read; every complete way is a path, in that order) on a graph cut from one sequence with branches, repeats and dead ends;
    a buffer too small for all records leaves the rest to the caller (state 3), never half a record."""
    import ctypes as C
    from shannon_amd import device, _lib
    rng = np.random.default_rng(seed)
    ctx = device.Context(0)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 4000))
    # nodes: consecutive pieces of the sequence overlapping by K - 1 bases (a path graph), plus variant pieces (one base changed) that
    # branch off and rejoin, plus short pieces (several hops per read)
    cuts = [0]
    while cuts[-1] < len(base) - 200:
        cuts.append(cuts[-1] + int(rng.integers(20, 120)))
    nodes, edges = [], {}                                            # edges[src] = [(dst, offset into dst at which dst continues src)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        nodes.append(base[a:b + K - 1])
    for i in range(len(nodes) - 1):
        edges.setdefault(i, []).append((i + 1, K - 1))
    n_main = len(nodes)
    for i in range(2, n_main - 2, 5):                                # a variant of node i: same ends, one base changed in the middle
        v = list(nodes[i]); j = len(v) // 2; v[j] = "ACGT"[("ACGT".index(v[j]) + 1) % 4]
        if len(v) > 2 * K:
            nodes.append("".join(v))
            edges.setdefault(i - 1, []).insert(0, (len(nodes) - 1, K - 1))     # (listed BEFORE the true successor)
            edges.setdefault(len(nodes) - 1, []).append((i + 1, K - 1))
    for i in range(4, n_main - 2, 7):                                # a second edge to the true successor with a wrong overlap: never agrees
        edges[i].append((i + 1, 3))
    L = 100
    reads = []
    for _ in range(3000):
        a = int(rng.integers(0, len(base) - L)); r = base[a:a + L]
        if rng.random() < 0.3:                                       # follow a variant
            r = list(r); j = int(rng.integers(0, L)); r[j] = "ACGT"[("ACGT".index(r[j]) + 1) % 4]; r = "".join(r)
        reads.append(r)

    def search(seq, so, node, i, hops, cur, out):
        nl = len(nodes[node]) - i
        cur.append(node)
        if hops <= 0 or len(seq) - so <= nl:
            out.append(list(cur)); cur.pop(); return
        so2 = so + nl
        for dst, ov in edges.get(node, []):
            n = min(len(seq) - so2, len(nodes[dst]) - ov)
            if ov <= len(nodes[dst]) and seq[so2:so2 + n] == nodes[dst][ov:ov + n]:
                search(seq, so2, dst, ov, hops - 1, cur, out)
        cur.pop()
    d = device.Reads.from_strings(ctx, reads)
    text = np.frombuffer("".join(nodes).encode(), np.uint8)
    off = np.zeros(len(nodes) + 1, np.uint64); off[1:] = np.cumsum([len(n) for n in nodes])
    eoff = np.zeros(len(nodes) + 1, np.uint32)
    edst, eov = [], []
    for i in range(len(nodes)):
        for dst, ov in edges.get(i, []):
            edst.append(dst); eov.append(ov)
        eoff[i + 1] = len(edst)
    edst, eov = np.asarray(edst + [0], np.uint32), np.asarray(eov + [0], np.uint32)
    state = np.empty(len(reads), np.uint8); node = np.empty(len(reads), np.int32); ofs = np.empty(len(reads), np.uint32)
    paths = np.zeros(cap, np.int32); used = C.c_uint64()
    _lib.check(_lib.lib().shn_known_paths_search(ctx.h, d.h, K, text.ctypes.data, off.ctypes.data, len(nodes), eoff.ctypes.data, edst.ctypes.data,
                                                 eov.ctypes.data, state.ctypes.data, node.ctypes.data, ofs.ctypes.data, paths.ctypes.data, cap, C.byref(used)))
    got = {}
    at = 0
    while at + 2 <= used.value and paths[at + 1] > 0:
        r, ln = int(paths[at]), int(paths[at + 1])
        got.setdefault(r, []).append(paths[at + 2:at + 2 + ln].tolist())
        at += 2 + ln
    n4 = n3 = 0
    for r, sq in enumerate(reads):
        if state[r] == 4:
            n4 += 1
            want = []
            search(sq, 0, int(node[r]), int(ofs[r]), 30, [], want)
            assert got.get(r, []) == want, (r, got.get(r), want)
        else:
            assert r not in got
            n3 += state[r] == 3
    assert n4 > 100 and any(len(p) > 1 for p in got.values()) and any(len(q) >= 3 for p in got.values() for q in p)
    assert (n3 == 0) == (cap >= 1 << 16)                             # with room every read of state 3 was searched there
    d.close(); ctx.close()
