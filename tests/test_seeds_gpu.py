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
