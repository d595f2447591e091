"""GPU parity: shn_count_k1mers (HIP) vs the oracle and vs golden vectors (row a2)."""
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


def oracle_table(records, k1):
    from oracle import count
    keys, cnts = count.count_k1mers_packed(records, k1)
    return keys, cnts.astype(np.uint64)


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_count_matches_golden(ctx, name):
    from shannon_amd import device
    from oracle import seqs
    g = load_case(name)
    inp = load_inputs(name)
    sets = [device.Reads.from_strings(ctx, r) for r in inp]
    t = count_case(ctx, name, sets)
    keys, cnts = t.dump(lower=1)
    assert len(keys) == g["n_k1mers"]
    assert int(cnts.astype(np.uint64).sum()) == g["k1mer_total"]
    rows = [[device.key_to_str(k, g["K"] + 1), int(c)] for k, c in zip(keys, cnts)]
    assert digest(rows) == g["k1mer_counts_digest"]      # bit-exact sorted multiset
    # and against the oracle restatement on the strand-doubled records
    dbl = read_files(name, inp)
    ok, oc = oracle_table([r for f in dbl for r in f], g["K"] + 1)
    assert np.array_equal(ok, keys) and np.array_equal(oc, cnts.astype(np.uint64))


@pytest.mark.parametrize("k1,L,n,canon", [(26, 100, 20000, True), (32, 100, 5000, True), (32, 100, 5000, False),
                                          (2, 40, 300, True), (25, 25, 1000, True), (21, 150, 3000, False),
                                          (26, 100, 300000, True)])
def test_count_random_vs_oracle(ctx, k1, L, n, canon):
    from shannon_amd import device
    from oracle import count
    rng = np.random.default_rng(k1 * 1000 + L)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    if L >= 40:
        codes[rng.random((n, L)) < 0.002] = 255          # sprinkle N
        codes[: n // 50] = codes[0]                       # heavy duplicates
        codes[n // 50: n // 25] = 3                       # poly-T (all-ones keys)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, codes)], k1, both_strands=canon)
    if canon:
        rc = np.where(codes[:, ::-1] == 255, 255, 3 - codes[:, ::-1]).astype(np.uint8)
        ok, oc = count.count_k1mers_matrix(np.concatenate([codes, rc]), k1)
    else:
        ok, oc = count.count_k1mers_matrix(codes, k1)
    keys, cnts = t.dump(lower=1)
    assert np.array_equal(ok, keys)
    assert np.array_equal(oc.astype(np.uint64), cnts.astype(np.uint64))
    # lookups through the bucket index
    raw_k, raw_c = t.download()
    sel = raw_k[:: max(1, len(raw_k) // 1000)]
    assert np.array_equal(t.lookup(sel), raw_c[:: max(1, len(raw_k) // 1000)])
    assert (t.lookup(np.array([0x1234567 % (1 << (2 * k1 - 1))], dtype=np.uint64)) >= 0).all()


def test_empty_and_short_reads(ctx):
    from shannon_amd import device
    t = device.count_k1mers(ctx, [device.Reads.from_strings(ctx, ["ACG", "", "ACGTN"])], 26)
    assert len(t) == 0 and t.total == 0
    t = device.count_k1mers(ctx, [device.Reads.from_strings(ctx, ["ACGTACGTAC", "NNNNNNNNNNNN", "acgtacgtacgt"])], 10, both_strands=False)
    keys, cnts = t.dump()
    assert dict(zip([device.key_to_str(k, 10) for k in keys], cnts.tolist())) == {"ACGTACGTAC": 2, "CGTACGTACG": 1, "GTACGTACGT": 1}


def test_pairs_and_shard_roundtrip(ctx):
    """The multi-GPU exchange pieces on one device: shard a table by owner hash, rebuild from the
    concatenated shards (duplicated once) -> counts double."""
    import torch
    from shannon_amd import device
    rng = np.random.default_rng(7)
    codes = rng.integers(0, 4, size=(20000, 100), dtype=np.uint8)
    t = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, codes)], 26)
    n = len(t)
    dk = torch.empty(2 * n, dtype=torch.int64, device="cuda:0")
    dc = torch.empty(2 * n, dtype=torch.int32, device="cuda:0")
    per = t.shard(8, dk.data_ptr(), dc.data_ptr())
    assert int(per.sum()) == n and per.min() > 0.8 * n / 8
    dk[n:] = dk[:n]
    dc[n:] = dc[:n]
    torch.cuda.synchronize()
    t2 = device.Table.from_pairs(ctx, dk.data_ptr(), dc.data_ptr(), 2 * n, 26, True)
    k1_, c1_ = t.dump()
    k2_, c2_ = t2.dump()
    assert np.array_equal(k1_, k2_) and np.array_equal(2 * c1_.astype(np.uint64), c2_.astype(np.uint64))
    assert t2.total == 2 * t.total


@pytest.mark.parametrize("both,with_n,log2,K1", [(True, False, None, 26), (False, True, None, 26), (True, True, 8, 26), (True, False, 12, 26),
                                                 (True, True, None, 32), (True, False, 10, 32)])
def test_one_pass_counting_equals_the_partition_pipeline(both, with_n, log2, K1, monkeypatch):
    """Large inputs are counted in one pass into a global hash table (LDS pre-aggregation + device-scope atomics) whose
    pairs then take the pairs path; forced here on a small input, also with a table that starts far too small and has to
    grow: the same table as the partition pipeline."""
    from shannon_amd import device
    rng = np.random.default_rng(11)
    base = rng.integers(0, 4, 3000, dtype=np.uint8)
    starts = rng.integers(0, len(base) - 100, 40000)
    reads = base[starts[:, None] + np.arange(100)]
    err = rng.random(reads.shape) < 0.01
    reads = np.where(err, (reads + rng.integers(1, 4, reads.shape)) & 3, reads).astype(np.uint8)
    reads[:50] = 3                                        # poly-T reads: at k1 = 32 the all-ones key, stored as its reverse complement
    reads[50:80] = 0
    if with_n:
        reads[rng.random(reads.shape) < 0.002] = 4
    ctx = device.Context(0)
    try:
        d = device.Reads.from_codes(ctx, reads)
        monkeypatch.setenv("SHN_COUNT_DIRECT", "0")
        t0 = device.count_k1mers(ctx, [d], K1, both)
        k0, c0 = t0.download()
        monkeypatch.setenv("SHN_COUNT_DIRECT", "2")
        if log2 is not None:
            monkeypatch.setenv("SHN_COUNT_DIRECT_LOG2", str(log2))
        t1 = device.count_k1mers(ctx, [d], K1, both)
        k1, c1 = t1.download()
        o0, o1 = np.argsort(k0), np.argsort(k1)
        assert np.array_equal(k0[o0], k1[o1]) and np.array_equal(c0[o0], c1[o1])
        assert len(t0) == len(t1) and int(c1.sum()) == int(c0.sum())
        t0.close(); t1.close(); d.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("chunk,ragged", [(700000, False), (123457, False), (250000, True)])
def test_chunked_counting_equals_one_pass_over_all_reads(chunk, ragged, monkeypatch):
    """Beyond 2^31 windows (BASELINE configs[2]: 7.5 G) the reads are counted chunk by chunk and the chunk tables are reduced by
    key; forced here with a small chunk, over two read sets, fixed-length and ragged reads, with N bases."""
    from shannon_amd import device
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, 5000, dtype=np.uint8)
    sets_np = []
    for n in (9000, 6000):
        starts = rng.integers(0, len(base) - 100, n)
        reads = base[starts[:, None] + np.arange(100)]
        err = rng.random(reads.shape) < 0.01
        reads = np.where(err, (reads + rng.integers(1, 4, reads.shape)) & 3, reads).astype(np.uint8)
        reads[rng.random(reads.shape) < 0.001] = 4
        sets_np.append(reads)
    ctx = device.Context(0)
    try:
        if ragged:
            A = np.frombuffer(b"ACGTN", np.uint8)
            ds = [device.Reads.from_strings(ctx, [A[r[: 60 + (i * 7) % 41]].tobytes().decode() for i, r in enumerate(m)]) for m in sets_np]
        else:
            ds = [device.Reads.from_codes(ctx, m) for m in sets_np]
        monkeypatch.setenv("SHN_COUNT_DIRECT", "0")
        t0 = device.count_k1mers(ctx, ds, 26, True)
        monkeypatch.setenv("SHN_COUNT_CHUNK", str(chunk))
        t1 = device.count_k1mers(ctx, ds, 26, True)
        k0, c0 = t0.download()
        k1, c1 = t1.download()
        o0, o1 = np.argsort(k0), np.argsort(k1)
        assert np.array_equal(k0[o0], k1[o1]) and np.array_equal(c0[o0], c1[o1]) and t0.total == t1.total
        for t in (t0, t1):
            t.close()
        for d in ds:
            d.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("K1,both,ragged,with_n,bits,pool,slots", [(26, True, False, False, None, None, None), (26, True, True, True, 6, None, None),
                                                                    (26, False, False, True, None, "0.0001", None), (20, True, False, False, 9, None, None),
                                                                    (31, True, True, True, None, None, None), (32, True, False, True, 5, None, None),
                                                                    (31, False, False, False, 12, "0.01", None), (23, False, True, True, 4, None, None),
                                                                    (26, True, False, True, 7, None, 11), (26, True, True, False, None, None, 12),
                                                                    (26, True, False, True, 8, None, -1), (32, True, True, True, None, "0.001", -1)])
def test_superkmer_counting_equals_the_partition_pipeline(K1, both, ragged, with_n, bits, pool, slots, monkeypatch):
    """Large diverse inputs are counted through super-k-mers (csrc/count_sk.hip: minimizer runs of a read's windows travel as 16-byte
    records, windows are expanded to keys in the bucket kernel's LDS, the buckets' pairs are reduced by the pairs path); forced
    here on small inputs -- fixed-length and ragged reads (some shorter than k1, some exactly k1), N bases, poly-A / poly-T reads,
    heavy duplicates, canonical and forward counting, k1 = 20 .. 32, several bucket grids, a pair pool that starts too small, reads
    with more records than slots (the overflow list), the buckets' pairs through the pairs path instead of straight into the table:
    the same table as the partition pipeline, and as the oracle."""
    from shannon_amd import device
    from oracle import count
    rng = np.random.default_rng(K1 * 7 + (3 if both else 0))
    base = rng.integers(0, 4, 4000, dtype=np.uint8)
    starts = rng.integers(0, len(base) - 100, 30000)
    reads = base[starts[:, None] + np.arange(100)]
    err = rng.random(reads.shape) < 0.01
    reads = np.where(err, (reads + rng.integers(1, 4, reads.shape)) & 3, reads).astype(np.uint8)
    reads[:40] = 3
    reads[40:70] = 0
    reads[70:400] = reads[70]                             # one read 330 times
    if with_n:
        reads[rng.random(reads.shape) < 0.002] = 4
    ctx = device.Context(0)
    try:
        if ragged:
            A = np.frombuffer(b"ACGTN", np.uint8)
            lens = [K1 - 1, K1, K1 + 1, 100, 47] + [int(x) for x in rng.integers(K1 - 3, 101, len(reads) - 5)]
            strings = [A[r[:n]].tobytes().decode() for r, n in zip(reads, lens)]
            d = device.Reads.from_strings(ctx, strings)
            recs = strings
        else:
            d = device.Reads.from_codes(ctx, reads)
            recs = None
        monkeypatch.setenv("SHN_COUNT_DIRECT", "0")
        monkeypatch.setenv("SHN_COUNT_SK", "0")
        t0 = device.count_k1mers(ctx, [d], K1, both)
        k0, c0 = t0.download()
        monkeypatch.setenv("SHN_COUNT_SK", "2")
        if bits is not None:
            monkeypatch.setenv("SHN_COUNT_SK_BITS", str(bits))
        if pool is not None:
            monkeypatch.setenv("SHN_COUNT_SK_POOL", pool)
        if slots is not None and slots > 0:
            monkeypatch.setenv("SHN_COUNT_SK_SLOTS", str(slots))
        if slots == -1:                                   # the pairs path behind the buckets (the fallback of a bucket no key range splits)
            monkeypatch.setenv("SHN_COUNT_SK_LAYOUT", "0")
        t1 = device.count_k1mers(ctx, [d], K1, both)
        assert "count.sk_buckets" in ctx.timers()
        k1, c1 = t1.download()
        o0, o1 = np.argsort(k0), np.argsort(k1)
        assert np.array_equal(k0[o0], k1[o1]) and np.array_equal(c0[o0], c1[o1]) and t0.total == t1.total and len(k1) > 1000
        if not ragged:
            codes = np.where(reads == 4, 255, reads).astype(np.uint8)
            if both:
                rc = np.where(codes[:, ::-1] == 255, 255, 3 - codes[:, ::-1]).astype(np.uint8)
                ok, oc = count.count_k1mers_matrix(np.concatenate([codes, rc]), K1)
                keys, cnts = t1.dump(lower=1)
            else:
                ok, oc = count.count_k1mers_matrix(codes, K1)
                keys, cnts = t1.dump(lower=1)
            assert np.array_equal(ok, keys) and np.array_equal(oc.astype(np.uint64), cnts.astype(np.uint64))
        t0.close(); t1.close(); d.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["se_K24", "pe_K25", "syn_pe_s20_K31", "syn_pe_ss_s69"])
def test_superkmer_counting_matches_golden(ctx, name, monkeypatch):
    """the reference's own k1-mer tables (golden fixtures) through the super-k-mer path"""
    from shannon_amd import device
    monkeypatch.setenv("SHN_COUNT_DIRECT", "0")
    monkeypatch.setenv("SHN_COUNT_SK", "2")
    g = load_case(name)
    inp = load_inputs(name)
    sets = [device.Reads.from_strings(ctx, r) for r in inp]
    t = count_case(ctx, name, sets)
    keys, cnts = t.dump(lower=1)
    assert len(keys) == g["n_k1mers"] and int(cnts.astype(np.uint64).sum()) == g["k1mer_total"]
    rows = [[device.key_to_str(k, g["K"] + 1), int(c)] for k, c in zip(keys, cnts)]
    assert digest(rows) == g["k1mer_counts_digest"]
