"""GPU parity at BASELINE configs[1] size (10 M synthetic 2x100 bp reads, K=25) through size-independent
properties -- the oracle cannot run at this size, the properties can be checked exactly:
  counting   : total = #windows; keys unique and canonical; linearity count(R1 u R2) = count(R1) + count(R2);
               a random sample of k1-mers agrees with a brute-force count over the raw reads
  extension  : every k1-mer is used at most once over all contigs (extension_correction.py:223-245, the global
               `traversed` set), every contig window is a counted k1-mer, the allowed dict is exactly those windows
  whole path : two runs give the same transcripts (the walk fixpoint is unique whatever the race outcomes)."""
import os, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_READS = 10_000_000
K = 25
K1 = K + 1
K1S = [26, 32]          # -K 25 (BASELINE) and -K 31 (64-bit keys, configs[4])


def _rc_keys(keys, k):
    out = np.zeros_like(keys)
    x = keys.copy()
    for _ in range(k):
        out = (out << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    return out


@pytest.fixture(scope="module")
def batch():
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from shannon_amd import device
    dev = torch.device("cuda", 0)
    r1, r2 = bench.gen_reads(N_READS // 2, 20240501, 1, dev)
    ctx = device.Context(0)
    d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
    yield ctx, r1, r2, d1, d2
    d1.close(); d2.close(); ctx.close()


def _brute_count(codes, key, k):
    """occurrences of the k-mer `key` and of its reverse complement as windows of the code matrix rows"""
    pat = np.array([(int(key) >> (2 * (k - 1 - j))) & 3 for j in range(k)], dtype=np.uint8)
    total = 0
    for p in (pat, (3 - pat)[::-1]):
        L = codes.shape[1]
        hit = np.ones((codes.shape[0], L - k + 1), dtype=bool)
        for j in range(k):
            hit &= codes[:, j:L - k + 1 + j] == p[j]
            if not hit.any():
                break
        total += int(hit.sum())
        if np.array_equal(p, (3 - p)[::-1]):          # palindrome: forward == reverse complement, count once
            break
    return total


@pytest.mark.parametrize("K1", K1S)
def test_fullsize_count_properties(batch, K1):
    from shannon_amd import device
    ctx, r1, r2, d1, d2 = batch
    W = 100 - K1 + 1
    t = device.count_k1mers(ctx, [d1, d2], K1, both_strands=True)
    ta = device.count_k1mers(ctx, [d1], K1, both_strands=True)
    tb = device.count_k1mers(ctx, [d2], K1, both_strands=True)
    try:
        assert t.total == N_READS * W
        keys, cnts = t.download()
        assert int(cnts.astype(np.uint64).sum()) == N_READS * W
        order = np.argsort(keys, kind="stable")
        ks, cs = keys[order], cnts[order]
        assert np.all(ks[1:] > ks[:-1])                                   # unique
        assert np.all(ks <= _rc_keys(ks, K1))                             # canonical representative
        # linearity: the table of both mates is the key-wise sum of the two tables
        ka, ca = ta.download()
        kb, cb = tb.download()
        allk = np.concatenate([ka, kb])
        allc = np.concatenate([ca, cb]).astype(np.uint64)
        o = np.argsort(allk, kind="stable")
        allk, allc = allk[o], allc[o]
        first = np.concatenate([[True], allk[1:] != allk[:-1]])
        summed = np.add.reduceat(allc, np.nonzero(first)[0])
        assert np.array_equal(allk[first], ks) and np.array_equal(summed, cs.astype(np.uint64))
        # brute force on a sample of keys (heavy and light) over a slice of the reads, against a table of that slice
        sl = 200_000
        sub = np.concatenate([r1[:sl], r2[:sl]])
        ts = device.count_k1mers(ctx, [device.Reads.from_codes(ctx, sub)], K1, both_strands=True)
        sk, sc = ts.download()
        rng = np.random.default_rng(5)
        pick = np.concatenate([np.argsort(sc)[-4:], rng.choice(len(sk), 8, replace=False)])
        for i in pick.tolist():
            assert _brute_count(sub, sk[i], K1) == int(sc[i])
        ts.close()
    finally:
        t.close(); ta.close(); tb.close()


@pytest.mark.parametrize("K1", K1S)
def test_fullsize_extension_properties(batch, K1):
    from shannon_amd import device, extension_correction as ec
    ctx, r1, r2, d1, d2 = batch
    t = device.count_k1mers(ctx, [d1, d2], K1, both_strands=True)
    try:
        res = ec.run_correction(ctx, t, 3, 75, 500)
        contigs = res.contigs
        assert len(contigs) > 100
        wins = np.concatenate([ec.windows_to_keys(c, K1) for c in contigs])
        assert len(np.unique(wins)) == len(wins)                          # each k1-mer string in at most one contig, once
        canon = np.minimum(wins, _rc_keys(wins, K1))
        w = t.lookup(canon)
        assert np.all(w >= 1)                                             # every window was counted
        # allowed dict = exactly those windows with their weights (extension_correction.py:366-369, 404-408)
        assert len(res.allowed) == len(wins)
    finally:
        t.close()


def test_fullsize_whole_path_deterministic(batch):
    from shannon_amd import pipeline, kmers_for_component as kfc
    ctx, r1, r2, d1, d2 = batch
    store = kfc.ReadStore(r1, r2)
    a = pipeline.assemble_resident(ctx, d1, d2, store, K=K, sample="full", seed=1)
    b = pipeline.assemble_resident(ctx, d1, d2, store, K=K, sample="full", seed=1)
    assert a.extension.contigs == b.extension.contigs
    assert a.final == b.final and len(a.final) > 50
    # every transcript is made of counted k1-mers only
    from shannon_amd import device, extension_correction as ec
    t = device.count_k1mers(ctx, [d1, d2], K1, both_strands=True)
    try:
        for name, seq in list(a.final.items())[:200]:
            ks = ec.windows_to_keys(seq, K1)
            assert np.all(t.lookup(np.minimum(ks, _rc_keys(ks, K1))) >= 1), name
    finally:
        t.close()
