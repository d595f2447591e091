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
