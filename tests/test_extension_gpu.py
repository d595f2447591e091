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
    t = count_case(ctx, name, sets)
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


@pytest.mark.parametrize("lost_round", [1, 3, 6, 12])
def test_fixpoint_audit_repairs_lost_marks(ctx, lost_round, monkeypatch, capfd):
    """The rounds' change tracking is an optimisation; the audit after the last block is the definition of the
    fixpoint.  Fault injection: every dirty mark of one round is dropped (blocks settle too early) -- the audit has to
    see it, reopen the blocks and end on the same walks.  SHN_EXT_AUDIT=2 re-derives every walk sequentially from the
    final claims (one thread per walk) and fails the call if one differs."""
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=78)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        monkeypatch.setenv("SHN_EXT_AUDIT", "2")
        e = ec.Extension(ctx, t, 3)
        ref = e.stats()
        e.close()
        assert "fixpoint audit" not in capfd.readouterr().err          # a healthy run never trips the audit
        monkeypatch.setenv("SHN_EXT_FAULT", str(lost_round))
        e = ec.Extension(ctx, t, 3)
        got = e.stats()
        e.close()
        assert "reopening all blocks" in capfd.readouterr().err
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)
    finally:
        t.close()


class _Abort(Exception):
    pass


def _run_virtual_ranks(world, run):
    """Emulates `world` ranks on one GPU: run(rank, gather) is executed rank after rank; the collectives are replayed --
    every pass completes one more collective for all ranks, until a pass runs through."""
    history = []                                      # completed collectives: ("gather", [obj per rank]) / ("max", value)
    while True:
        pending, results = {}, {}
        for rank in range(world):
            calls = [0]

            class G(object):
                pass
            g = G()
            g.world, g.rank = world, rank

            def collective(kind, obj, rank=rank, calls=calls):
                k = calls[0]
                calls[0] += 1
                if k < len(history):
                    assert history[k][0] == kind
                    return history[k][1]
                pending[rank] = (kind, obj)
                raise _Abort()
            g.all_gather = lambda obj, c=collective: c("gather", obj)
            g.all_reduce_max = lambda v, c=collective: c("max", v)
            try:
                results[rank] = run(rank, g)
            except _Abort:
                pass
        if len(results) == world:
            return results, history
        assert len(pending) == world and len({k for k, _ in pending.values()}) == 1      # every rank is at the same collective
        kind = pending[0][0]
        history.append((kind, [pending[r][1] for r in range(world)] if kind == "gather" else max(pending[r][1] for r in range(world))))


@pytest.mark.parametrize("world,limit_override", [(2, None), (5, None), (3, -1)])
def test_component_sharded_contig_stages_equal_the_global_pass(ctx, world, limit_override, monkeypatch):
    """Walks and contig stages sharded by component (duplicate_check per shard + GPU join against the other shards'
    accepted contigs) give the contigs, connections and components of the unsharded run -- also when the guard trips
    and every rank falls back to the global pass (limit_override = -1 forces that)."""
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=80)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        if limit_override is not None:
            monkeypatch.setattr(ec, "foreign_interference", lambda *a, **k: 1)
        results, history = _run_virtual_ranks(world, lambda rank, g: ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False,
                                                                                       shard=(world, rank), gather=g))
        assert history[1] == ("max", 0 if limit_override is None else 1)
        for rank in range(world):
            res = results[rank]
            assert res.contigs == ref.contigs
            assert res.connections == ref.connections and [list(v) for v in res.connections.values()] == [list(v) for v in ref.connections.values()]
            assert res.components == ref.components
            assert res.single_contigs == ref.single_contigs and res.remaining == ref.remaining and res.big_components == ref.big_components
    finally:
        t.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_contig_stage_keeps_connections_through_shared_end_kmers(ctx, world):
    """Unrelated genes whose transcripts END (or START) in the same K-mer: their last k1-mers x.m / x'.m are not adjacent and have
    no common successor, yet contig_connections joins the contigs through the shared K-mer m (extension_correction.py:372-390).
    The component labelling unites such sibling k1-mers, so the contigs land on one rank and the sharded result keeps the edge."""
    from shannon_amd import device, synth, extension_correction as ec
    rng = np.random.Generator(np.random.PCG64(123))
    tail, head = rng.integers(0, 4, 25, dtype=np.uint8), rng.integers(0, 4, 25, dtype=np.uint8)
    isos = []
    for gix in range(12):
        body = rng.integers(0, 4, int(rng.integers(420, 700)), dtype=np.uint8)
        if gix < 4:
            body = np.concatenate([body, tail])                 # four genes end in the same 25-mer
        elif gix < 7:
            body = np.concatenate([head, body])                 # three start with the same 25-mer
        isos.append(body)
    r1, r2 = synth.sample_pairs(isos, 40000, 7, err=0.002)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        shared = [a for a, d in ref.connections.items() if d]
        assert len(shared) >= 4                                  # the end-sharing contigs are connected in the unsharded run
        results, history = _run_virtual_ranks(world, lambda rank, g: ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False,
                                                                                       shard=(world, rank), gather=g))
        for rank in range(world):
            res = results[rank]
            assert res.contigs == ref.contigs
            assert res.connections == ref.connections and [list(v) for v in res.connections.values()] == [list(v) for v in ref.connections.values()]
            assert res.components == ref.components and res.single_contigs == ref.single_contigs and res.remaining == ref.remaining
    finally:
        t.close()


def test_rmer_join_counts_like_duplicate_check(ctx):
    from shannon_amd import extension_correction as ec
    rng = np.random.default_rng(3)
    A = "ACGT"
    contigs = ["".join(A[c] for c in rng.integers(0, 4, n)) for n in (120, 200, 90)]
    contigs.append(contigs[1] + "A" + contigs[1])                                     # every occurrence counts
    cand = [contigs[0][10:70] + "".join(A[c] for c in rng.integers(0, 4, 40)),
            "".join(A[c] for c in rng.integers(0, 4, 100)),
            contigs[2][:40] + contigs[1][50:80]]
    hc, hs, hf = ec.rmer_join(ctx, cand, contigs, 15)
    got = sorted(zip(hc.tolist(), hs.tolist(), hf.tolist()))
    ref = []
    for ci, c in enumerate(cand):
        for i in range(len(c) - 14):
            for di, d in enumerate(contigs):
                for j in range(len(d) - 14):
                    if d[j:j + 15] == c[i:i + 15]:
                        ref.append((ci, i, di))
    assert got == sorted(ref) and len(ref) > 100
    assert len(ec.rmer_join(ctx, [], contigs, 15)[0]) == 0 and len(ec.rmer_join(ctx, cand, [], 15)[0]) == 0


def test_foreign_interference_rules(ctx):
    """the guard of the sharded duplicate check on hand-made cases (r = 15, f = 0.5)"""
    from shannon_amd import extension_correction as ec
    rng = np.random.default_rng(4)
    A = "ACGT"
    rnd = lambda n: "".join(A[c] for c in rng.integers(0, 4, n))
    d = rnd(300)
    # accepted candidate that a foreign earlier contig covers by more than half -> unsafe; by less than half -> safe
    other = lambda b: A[(A.index(b) + 1) % 4]        # a base that breaks the match right after the shared stretch
    c_big = d[20:100] + other(d[100]) + rnd(39)      # 80 of 120 bases shared
    c_small = d[20:45] + other(d[45]) + rnd(94)      # 25 of 120 bases shared (11 windows)
    local = [(50, 5, c_big), (40, 6, c_small)]
    foreign_early, foreign_late = [(90, 1, d)], [(10, 1, d)]
    assert ec.foreign_interference(ctx, local, [1, 2], [0, 0], foreign_early) == 1
    assert ec.foreign_interference(ctx, local[1:], [1], [0], foreign_early) == 0
    assert ec.foreign_interference(ctx, local, [1, 2], [0, 0], foreign_late) == 0        # later in the walk order: irrelevant
    # rejected candidate: safe while the foreign contig has fewer hits than its own best contig
    assert ec.foreign_interference(ctx, local[1:], [0], [12], foreign_early) == 0       # 11 foreign hits < 12
    assert ec.foreign_interference(ctx, local[1:], [0], [11], foreign_early) == 1       # tie: the foreign one could be the last


@pytest.mark.parametrize("lost_round", [1, 3, 7])
def test_pipelined_contig_stage_survives_a_reopened_fixpoint(ctx, lost_round, monkeypatch, capfd):
    """The contig stage runs beside the walks on the blocks that are already final.  If the fixpoint audit then reopens the
    blocks (forced here by dropping one round's marks), what was handed over is void: the result must still be the one of
    the plain, unpipelined run."""
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=81)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        monkeypatch.setenv("SHN_EXT_PIPELINE", "0")
        ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        monkeypatch.setenv("SHN_EXT_PIPELINE", "1")
        T = {}
        got = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, timings=T)
        assert "ext.contig_graph (beside the walks)" in T                      # the pipelined path was taken
        assert got.contigs == ref.contigs and got.connections == ref.connections and got.components == ref.components
        capfd.readouterr()
        monkeypatch.setenv("SHN_EXT_FAULT", str(lost_round))
        T = {}
        got = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False, timings=T)
        assert "reopening all blocks" in capfd.readouterr().err
        assert "ext.contig_graph (beside the walks)" not in T                  # ... and abandoned
        assert got.contigs == ref.contigs and got.connections == ref.connections and got.components == ref.components
    finally:
        t.close()


def _variants(contigs, seed=5):
    """accepted contigs + near-copies, substrings and chimeras of them: clusters of several candidates, duplicates to
    reject, contigs that share K-mers at their ends"""
    rng = np.random.default_rng(seed)
    out = []
    for i, c in enumerate(contigs):
        out.append(c)
        if len(c) > 120:
            out.append(c[10:-10])
            mid = len(c) // 2
            out.append(c[:mid] + "".join("ACGT"[j] for j in rng.integers(0, 4, 40)) + c[mid:])
        if i and len(c) > 60 and len(contigs[i - 1]) > 60:
            out.append(contigs[i - 1][-50:] + c[:50] + "".join("ACGT"[j] for j in rng.integers(0, 4, 30)))
    return out


@pytest.mark.parametrize("name", sorted(MANIFEST))
@pytest.mark.parametrize("block0,pair_log2", [(None, None), (7, None), (1, 4), (-7, None), (1000, -2)])
def test_gpu_contig_stage_equals_the_sequential_stage(ctx, name, block0, pair_log2, monkeypatch):
    """shn_contig_stage (duplicate_check as block-wise rounds over one device sort of the r-mers, K-mer join on the GPU) ==
    shn_cgraph (the reference's sequential loop, checked against the oracle on the CPU): accepted flags, best-hit counts,
    connections with weights and insertion order.  Also with tiny blocks and a pair table that has to grow; -7: blocks of 7 with
    every candidate of a block evaluated in every round (the default re-evaluates only those behind a changed decision)."""
    from shannon_amd import extension_correction as ec
    if pair_log2 is not None and pair_log2 < 0:                      # (1000, -2): large blocks given up after 2 rounds and halved
        monkeypatch.setenv("SHN_CONTIG_MAX_ROUNDS", str(-pair_log2))
        pair_log2 = None
    if block0 is not None and block0 < 0:
        monkeypatch.setenv("SHN_CONTIG_INCREMENTAL", "0")
        block0 = -block0
    if block0 is not None:
        monkeypatch.setenv("SHN_CONTIG_BLOCK0", str(block0))
    if pair_log2 is not None:
        monkeypatch.setenv("SHN_CONTIG_PAIR_LOG2", str(pair_log2))
    g = load_case(name)
    for cands in (g["contigs"], _variants(g["contigs"]), _variants(g["contigs"], 9)[::-1]):
        if not cands:
            continue
        acc, coff, cnb, cw, best = ec.contig_stage(cands, g["K"] + 1, want_best=True)
        buf = np.frombuffer("".join(cands).encode(), np.uint8)
        offs = np.zeros(len(cands) + 1, np.uint64)
        offs[1:] = np.cumsum([len(c) for c in cands])
        acc2, best2, coff2, cnb2, cw2 = ec.contig_stage_gpu(ctx, buf, offs, g["K"] + 1)
        assert np.array_equal(acc, acc2) and np.array_equal(best, best2)
        assert coff2.tolist() == coff and cnb2.tolist() == cnb and cw2.tolist() == cw


@pytest.mark.parametrize("env", [{"SHN_CONTIG_PACKED": "0"}, {"SHN_CONTIG_SHARED": "0"}, {"SHN_CONTIG_PACKED": "0", "SHN_CONTIG_BLOCK0": "5"}])
def test_contig_stage_sorts_pairs_or_packed_words_alike(ctx, env, monkeypatch):
    """the r-mer index of duplicate_check comes from a sort of (r-mer << 32 | base index) words (round 6: 8 bytes per entry and
    pass); SHN_CONTIG_PACKED=0 sorts (key, value) pairs as before, SHN_CONTIG_SHARED=0 keeps every window in the index: the
    sequential stage's answers in all three forms"""
    from shannon_amd import extension_correction as ec
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name in sorted(MANIFEST)[:4]:
        g = load_case(name)
        for cands in (_variants(g["contigs"]), _variants(g["contigs"], 9)[::-1], g["contigs"][:1]):
            if not cands:
                continue
            acc, coff, cnb, cw, best = ec.contig_stage(cands, g["K"] + 1, want_best=True)
            buf = np.frombuffer("".join(cands).encode(), np.uint8)
            offs = np.zeros(len(cands) + 1, np.uint64)
            offs[1:] = np.cumsum([len(c) for c in cands])
            acc2, best2, coff2, cnb2, cw2 = ec.contig_stage_gpu(ctx, buf, offs, g["K"] + 1)
            assert np.array_equal(acc, acc2) and np.array_equal(best, best2)
            assert coff2.tolist() == coff and cnb2.tolist() == cnb and cw2.tolist() == cw


def test_gpu_contig_stage_on_many_genes(ctx):
    """300 genes: a few thousand candidates in many small clusters, through run_correction both ways"""
    from shannon_amd import device, synth, extension_correction as ec
    import os
    (r1, r2), _ = synth.make_dataset(150000, 300, seed=21)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)], 26)
    os.environ["SHN_CONTIG_GPU"] = "0"
    try:
        ref = ec.run_correction(ctx, t, 3, 75, 50)
        os.environ["SHN_CONTIG_GPU"] = "1"
        got = ec.run_correction(ctx, t, 3, 75, 50)
    finally:
        del os.environ["SHN_CONTIG_GPU"]
    assert len(ref.contigs) > 500
    assert got.contigs == ref.contigs and got.connections == ref.connections
    assert [list(v) for v in got.connections.values()] == [list(v) for v in ref.connections.values()]
    assert got.components == ref.components and got.single_contigs == ref.single_contigs
    assert got.remaining == ref.remaining and got.big_components == ref.big_components and got.allowed == ref.allowed


@pytest.mark.parametrize("env", [{"SHN_EXT_BULK": "1"}, {"SHN_EXT_BULK": "1", "SHN_EXT_DENSE": "1"}, {"SHN_EXT_BULK": "1", "SHN_EXT_DENSE": "1000000000"},
                                 {"SHN_EXT_BULK": "1", "SHN_EXT_PROMOTE_BULK": "0"}, {"SHN_EXT_BULK": "1", "SHN_EXT_PROMOTE_BULK": "1", "SHN_EXT_RESUME_WAVES": "3"},
                                 {"SHN_EXT_BULK": "1", "SHN_EXT_PROMOTE_BULK": "5", "SHN_EXT_DENSE": "1"}, {"SHN_EXT_BULK": "1", "SHN_EXT_PREPASS": "0"},
                                 {"SHN_EXT_PREPASS": "0"}, {"SHN_EXT_MEMO_RELEASE": "0"}, {"SHN_EXT_MEMO_RELEASE_MAX": "3"}, {"SHN_EXT_FIRST_LOOK": "1", "SHN_EXT_BULK": "1"},
                                 {"SHN_EXT_BULK": "1", "SHN_EXT_FRESH_SPLIT": "5", "SHN_EXT_FRESH_SPLIT_MIN": "32"},
                                 {"SHN_EXT_BULK": "1", "SHN_EXT_FRESH_SPLIT": "3", "SHN_EXT_FRESH_SPLIT_MIN": "32", "SHN_EXT_DENSE": "1"},
                                 # claim logs (round 6): off; with every round a bulk round releasing through the logs; with a pool that
                                 # runs out after a few hundred chunks (void logs: those rounds fall back to the begin pass); with rounds that
                                 # may give back at most 50 claims through memos / logs
                                 {"SHN_EXT_LOGS": "0"}, {"SHN_EXT_BULK": "1", "SHN_EXT_DENSE": "1000000000", "SHN_EXT_LOGS": "1"},
                                 {"SHN_EXT_BULK": "1", "SHN_EXT_DENSE": "1000000000", "SHN_EXT_LOG_CHUNKS": "256"},
                                 {"SHN_EXT_BULK": "1", "SHN_EXT_LOG_CHUNKS": "64"}, {"SHN_EXT_BULK": "2000", "SHN_EXT_TARGETED_MAX": "50"},
                                 {"SHN_EXT_BULK": "2000", "SHN_EXT_DENSE": "1000000000"}])
def test_bulk_rounds_give_the_same_contigs(ctx, env, monkeypatch):
    """Every round as a bulk round (thread walker only, no snapshot reads in a block's first round), with the begin / mark passes
    as they come, all dense or all following the line flags; the hand-over of long walks to the packed second launch
    (ext_walk_resume_kernel) off, after one step with three wavefronts for all of them, after five; the settling of void walks in a
    block's first round off; the release of re-run walks' claims by the streaming begin pass only / through memos only in rounds of
    at most three walks; the first look at the candidates' claims; a block's first round in 5 / 3 rank-ordered sub-launches: the same
    walks, contigs and connections as the default path of a small table."""
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(60000, 20, seed=91)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        ref = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("SHN_EXT_AUDIT", "2")
        got = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        assert got.contigs == ref.contigs and got.connections == ref.connections and len(ref.contigs) > 20
    finally:
        t.close()


def test_accept_filter_on_the_device_equals_the_reference_rule(ctx):
    """shn_ext_accept against extension_correction.py:361 evaluated with math.pow per walk (and against the numpy form): every
    non-void walk of a small run, thresholds chosen so that candidates fall on both sides and some exactly on the threshold"""
    import math
    from shannon_amd import device, synth, extension_correction as ec
    (r1, r2), _ = synth.make_dataset(40000, 12, seed=17)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, np.concatenate([r1, r2]))], 26)
    try:
        ext = ec.Extension(ctx, t, 3)
        live, nr, nl, tw = ext.live_stats(0)
        assert len(live) > 200
        for min_length, min_weight in ((75, 3), (40, 2), (120, 9), (27, 1)):
            thr = 2 * min_length * math.pow(min_weight, 0.25)
            want = [(int(r), 26 + int(a) + int(b)) for r, a, b, w in zip(live.tolist(), nr.tolist(), nl.tolist(), tw.tolist())
                    if 26 + a + b >= min_length and (26 + a + b) * math.pow(float(w) / max(1, a + b + 1), 0.25) >= thr]
            got_r, got_l = ext.accept(26, min_length, min_weight)
            assert list(zip(got_r.tolist(), got_l.tolist())) == want
            ref_r, ref_l = ec.accept_filter(live, nr, nl, tw, 26, min_length, min_weight, arrays=True)
            assert got_r.tolist() == ref_r.tolist() and got_l.tolist() == ref_l.tolist()
        ext.close()
    finally:
        t.close()
