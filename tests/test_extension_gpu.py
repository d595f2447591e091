"""GPU parity: contig extension (rows a3-a7) -- HIP walk fixpoint + host bookkeeping vs the
oracle and the reference's golden vectors."""
import numpy as np
import pytest
from golden_util import *

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from shannon_amd import device
    c = device.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_extension_matches_golden(ctx, name):
    from shannon_amd import device, extension_correction as ec
    g = load_case(name)
    psize = MANIFEST[name].get("partition_size", 500)
    sets = [device.Reads.from_strings(ctx, r) for r in load_inputs(name)]
    t = device.count_k1mers(ctx, sets, g["K"] + 1)
    res = ec.run_correction(ctx, t, 3, 75, psize)
    assert res.contigs == g["contigs"]                      # ordered list, bit-exact
    assert len(res.allowed) == g["n_allowed"]
    assert digest(sorted([k, v] for k, v in res.allowed.items())) == g["allowed_digest"]
    assert "".join(">Single_%d\n%s\n" % (i, c) for i, c in enumerate(res.single_contigs)) == g["single_contigs_fasta"]
    assert res.remaining == g["remaining"]
    assert [b[0] for b in res.big_components] == [b["contigs"] for b in g["big_components"]]
    assert [b[1] for b in res.big_components] == [b["metis"] for b in g["big_components"]]
    assert 1 <= res.iterations < 200


def test_all_walks_match_oracle_sequential(ctx):
    """Every walk (accepted or not) equals the sequential greedy of the oracle, on a noisy set."""
    from shannon_amd import device, synth, extension_correction as ec
    from oracle import count, extension
    (r1, r2), _ = synth.make_dataset(4000, 2, seed=11)
    codes = np.concatenate([r1, r2])
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, codes)], 26)
    ext = ec.Extension(ctx, t, 3)
    nr, nl, tw = ext.stats()
    live = np.nonzero(nr != ec.UNCLAIMED)[0]
    lens = 26 + nr[live].astype(np.int64) + nl[live].astype(np.int64)
    mine = ext.emit(live, lens)
    keys, cnts = t.dump()
    tab = {device.key_to_str(k, 26): int(c) for k, c in zip(keys, cnts)}
    kmers, k1 = extension.load_kmers([(k, tab[k]) for k in sorted(tab, reverse=True)])
    heaviest = sorted(kmers.items(), key=lambda kv: kv[1])
    traversed, ref, refw = set(), [], []
    while heaviest:
        s, w = heaviest.pop()
        if w < 3:
            break
        if s in traversed:
            continue
        traversed.add(s)
        r, rw, _ = extension._extend(s, True, traversed, kmers, k1)
        l, lw, _ = extension._extend(s, False, traversed, kmers, k1)
        ref.append("".join(reversed(l)) + s + "".join(r))
        refw.append(int(rw + lw + kmers[s]))
    assert mine == ref
    assert tw[live].tolist() == refw


def test_multigene_walks_match_oracle_and_are_deterministic(ctx):
    """A richer case (20 genes, ~200x coverage, errors): every walk equals the oracle's sequential greedy,
    the worklist fixpoint gives the same answer run after run, and emitted contigs are stable."""
    from shannon_amd import device, synth, extension_correction as ec
    from oracle import count, extension
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=77)
    codes = np.concatenate([r1, r2])
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, codes)], 26)
    outs = []
    for _ in range(2):
        ext = ec.Extension(ctx, t, 3)
        nr, nl, tw = ext.stats()
        live = np.nonzero(nr != ec.UNCLAIMED)[0]
        lens = 26 + nr[live].astype(np.int64) + nl[live].astype(np.int64)
        outs.append((ext.emit(live, lens), tw[live].tolist()))
        ext.close()
    assert outs[0] == outs[1]
    keys, cnts = t.dump()
    tab = {device.key_to_str(k, 26): int(c) for k, c in zip(keys, cnts)}
    kmers, k1 = extension.load_kmers([(k, tab[k]) for k in sorted(tab, reverse=True)])
    heaviest = sorted(kmers.items(), key=lambda kv: kv[1])
    traversed, ref, refw = set(), [], []
    while heaviest:
        s, w = heaviest.pop()
        if w < 3:
            break
        if s in traversed:
            continue
        traversed.add(s)
        r, rw, _ = extension._extend(s, True, traversed, kmers, k1)
        l, lw, _ = extension._extend(s, False, traversed, kmers, k1)
        ref.append("".join(reversed(l)) + s + "".join(r))
        refw.append(int(rw + lw + kmers[s]))
    assert outs[0][0] == ref
    assert outs[0][1] == refw


@pytest.mark.parametrize("world", [2, 3, 7])
def test_component_sharded_walks_union_is_the_unsharded_result(ctx, world):
    """shn_extend_sharded: the connected components of the k1-mer graph are dealt to `world` ranks; the candidates of
    all ranks, merged in the global walk order, are exactly the unsharded candidates, and so is everything after."""
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=78)
    codes = np.concatenate([r1, r2])
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, codes)], 26)
    try:
        ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        whole = []
        ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, shard=(1, 0), merge=lambda loc: (whole.extend(loc), list(loc))[1])
        assert whole == sorted(whole, key=lambda c: (-c[0], c[1])) and len(whole) > 20
        pieces = []
        for rank in range(world):
            ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, shard=(world, rank),
                              merge=lambda loc: (pieces.append(list(loc)), list(loc))[1])
        assert sum(1 for p in pieces if p) >= 2                      # the work really is spread
        merged = sorted((c for p in pieces for c in p), key=lambda c: (-c[0], c[1]))
        assert merged == whole
        # and a rank that merges everybody's candidates finishes exactly like the unsharded run
        res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, shard=(world, 0), merge=lambda loc: list(merged))
        assert res.contigs == ref.contigs and res.connections == ref.connections
        assert res.single_contigs == ref.single_contigs and res.remaining == ref.remaining
    finally:
        t.close()


@pytest.mark.parametrize("words", [256, 5000, 200000])
def test_full_memo_pool_only_costs_time(ctx, words, monkeypatch):
    """Memos are speculation: with a memo pool too small to hold them the walks are the same."""
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=79)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        monkeypatch.setenv("SHN_EXT_POOL_WORDS", str(words))
        got = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        assert got.contigs == ref.contigs and got.connections == ref.connections
    finally:
        t.close()
