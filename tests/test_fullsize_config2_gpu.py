"""BASELINE configs[2] at its full size on one GPU -- 100 M synthetic 2x100 bp reads of 20,000 genes, k = 25, --partition 500 -- and
the one-GPU slice of configs[4] (bench.py --config 4s: 100 M reads, -K 31 i.e. 64-bit k1-mers, 4,000 genes with exons up to 5 kb),
both through the WHOLE path, checked through properties that do not need an oracle run of that size (the oracle takes days there): conservation and
linearity of the counts, a sub-batch against the C restatement, exclusivity of the contigs' k1-mers, partition cover and bin
sizes, the routing rule on sampled reads (both directions), recovery of the planted transcripts, and run-to-run identity of the
whole output.  The batch is bench.py's (same generator, same seed): the digest of the transcripts is the one its JSON line prints."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

N_PAIRS, SEED = 50_000_000, 20240501
CFG = {"2": dict(K=25, genes=20000, exon_len=(80, 600), precision=0.99, recall=0.93, exact=0.06, routed=0.5),
       "4s": dict(K=31, genes=4000, exon_len=(80, 5000), precision=0.99, recall=0.80, exact=0.02, routed=0.3)}
DIGESTS = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")      # transcripts_sha256_16 of bench.py's line per config
_RC = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s[::-1].translate(_RC)


def canon_keys(keys, k):
    """canonical form of packed k-mers (2 bits per base, first base in the high bits): complement, then the 2-bit groups reversed
    by swapping neighbours, nibbles and bytes"""
    keys = keys.astype(np.uint64)
    x = ~keys
    x = ((x >> np.uint64(2)) & np.uint64(0x3333333333333333)) | ((x & np.uint64(0x3333333333333333)) << np.uint64(2))
    x = ((x >> np.uint64(4)) & np.uint64(0x0F0F0F0F0F0F0F0F)) | ((x & np.uint64(0x0F0F0F0F0F0F0F0F)) << np.uint64(4))
    x = x.byteswap() >> np.uint64(64 - 2 * k)
    return np.minimum(keys, x)


class Full(object):
    pass


@pytest.fixture(scope="module", params=["2", "4s"])
def full(request):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from shannon_amd import device, pipeline, kmers_for_component as kfc
    F = Full()
    F.cfg = request.param
    F.c = CFG[F.cfg]
    K = F.K = F.c["K"]
    F.K1 = K + 1
    dev = torch.device("cuda", 0)
    F.r1, F.r2 = bench.gen_reads(N_PAIRS, SEED, F.c["genes"], dev, read_seed=SEED + 2, exon_len=F.c["exon_len"])
    torch.cuda.empty_cache()
    F.ctx = device.Context(0)
    F.d1, F.d2 = device.Reads.from_codes(F.ctx, F.r1), device.Reads.from_codes(F.ctx, F.r2)
    F.store = kfc.ReadStore(F.r1, F.r2)
    F.run = lambda **kw: pipeline.assemble_resident(F.ctx, F.d1, F.d2, F.store, K=K, sample="bench", seed=1, **kw)
    F.R = F.run(keep_partitioning=True)
    F.sha = bench._final_sha(F.R.final)
    yield F
    F.d1.close(); F.d2.close()
    F.ctx.close()
    del F.r1, F.r2, F.store, F.R
    torch.cuda.empty_cache()


def test_counts_are_conserved_and_additive(full):
    """Every window of every read is counted once (no read holds an N): the counts sum to reads x (L - k1 + 1); and the table of
    the batch is the sum of the tables of its halves, key by key (100,000 sampled keys + keys absent from one half)."""
    from shannon_amd import device
    F = full
    K1 = F.K1
    L = F.r1.shape[1]
    t = device.count_k1mers(F.ctx, [F.d1, F.d2], K1, both_strands=True)
    assert t.total == 2 * N_PAIRS * (L - K1 + 1) == F.R.n_windows
    assert len(t) == F.R.n_k1mers
    keys, cnts = t.download()
    assert int(cnts.astype(np.uint64).sum()) == t.total               # canonical table: one entry per strand pair
    h = N_PAIRS // 2
    halves = []
    for sl in (slice(0, h), slice(h, N_PAIRS)):
        a, b = device.Reads.from_codes(F.ctx, F.r1[sl]), device.Reads.from_codes(F.ctx, F.r2[sl])
        halves.append(device.count_k1mers(F.ctx, [a, b], K1, both_strands=True))
        a.close(); b.close()
    assert halves[0].total + halves[1].total == t.total
    sel = keys[:: max(1, len(keys) // 100000)]
    assert np.array_equal(t.lookup(sel).astype(np.uint64), halves[0].lookup(sel).astype(np.uint64) + halves[1].lookup(sel).astype(np.uint64))
    k0, c0 = halves[0].download()
    sel0 = k0[:: max(1, len(k0) // 100000)]
    assert np.array_equal(t.lookup(sel0).astype(np.uint64), c0[:: max(1, len(k0) // 100000)].astype(np.uint64) + halves[1].lookup(sel0).astype(np.uint64))
    for x in halves:
        x.close()
    t.close()


def test_a_sub_batch_counts_like_the_c_restatement(full):
    """the first 250,000 pairs through the device counter and through oracle/count_c.c: the same table"""
    from shannon_amd import device
    from oracle import build_c
    F = full
    K1 = F.K1
    n = 250000
    build_c.build()
    codes = np.concatenate([F.r1[:n], F.r2[:n]])
    ok, oc, nw = build_c.count_canonical(codes, K1, True)
    a, b = device.Reads.from_codes(F.ctx, F.r1[:n]), device.Reads.from_codes(F.ctx, F.r2[:n])
    t = device.count_k1mers(F.ctx, [a, b], K1, both_strands=True)
    keys, cnts = t.download()
    o = np.argsort(keys, kind="stable")
    assert t.total == nw and np.array_equal(keys[o], ok) and np.array_equal(cnts[o].astype(np.uint64), oc.astype(np.uint64))
    t.close(); a.close(); b.close()


def test_contigs_own_their_k1mers(full):
    """Walks claim k1-mers exclusively (extension_correction.py:223-245: a traversed k1-mer string is never taken again): no k1-mer
    occurs twice among the contigs.  The two strands are walked as separate strings (the dictionary holds both), so a
    canonical k1-mer occurs at most twice."""
    from shannon_amd import _lib
    F = full
    K1 = F.K1
    contigs = F.R.extension.contigs
    assert len(contigs) > 20000 and min(len(c) for c in contigs) >= 75
    keys, _rows, nwin = _lib.string_windows(contigs, K1, want_keys=True)
    assert len(np.unique(keys)) == len(keys) == int(nwin.sum())
    _u, mult = np.unique(canon_keys(keys, K1), return_counts=True)
    assert int(mult.max()) <= 2


def test_partitions_cover_the_contigs_once(full):
    """Every contig of a multi-contig component lies in exactly one partition; the bins of small components close as soon as they
    exceed --partition (extension_correction.py:436-452), so a bin holds at most partition + largest small component contigs."""
    F = full
    P = F.R.partitioning
    ext = F.R.extension
    parts = {nm: cs for nm, cs in P["new_components"].items() if not nm.startswith("r2_")}
    placed = [c for cs in parts.values() for c in cs]
    assert len(placed) == len(set(placed)) == len(ext.contigs) - len(ext.single_contigs)
    assert set(placed) | set(ext.single_contigs) == set(ext.contigs)
    sizes = np.diff(np.asarray(ext.comp_off))
    small_max = int(sizes[sizes <= 500].max())
    for nm, cs in parts.items():
        if nm.startswith("cremaining"):
            assert len(cs) <= 500 + small_max
    closed = [len(cs) for nm, cs in parts.items() if nm.startswith("cremaining")][:-1]
    assert all(n > 500 for n in closed)
    # a big component is split into ceil-sized parts within gpmetis' balance bound (ufactor = 1000 x (overload - 1))
    for i, (contigs, _m) in enumerate(ext.big_components):
        ps = [len(cs) for nm, cs in parts.items() if nm.startswith("c%d_" % (i + 1))]
        assert sum(ps) == len(contigs) and max(ps) <= 2.0 * len(contigs) / len(ps) + 1
    assert list(F.R.partitions) == list(P["new_components"])


def test_routed_reads_hit_their_partition_and_only_they_do(full):
    """kmers_for_component.py:186-205, 358-403: a pair belongs to a partition iff one of its probes (the k1-windows at 0, k1, 2 k1,
    ... and the last one, of either mate) is a k1-mer of one of the partition's contigs.  Checked on the largest, a middle and
    the smallest partition: 2,000 routed pairs each, and 20,000 pairs drawn from the whole batch (strand-doubled numbering)."""
    F = full
    K1 = F.K1
    P = F.R.partitioning
    names = sorted(P["routes"], key=lambda nm: len(P["routes"][nm]))
    rng = np.random.default_rng(7)
    L = F.r1.shape[1]
    starts = list(range(0, L - K1, K1)) + [L - K1]
    for nm in (names[-1], names[len(names) // 2], names[0]):
        kset = set()
        for c in P["new_components"][nm]:
            for i in range(len(c) - K1 + 1):
                kset.add(c[i:i + K1])
        idx = np.asarray(P["routes"][nm])
        assert (np.diff(idx.astype(np.int64)) > 0).all()                  # file order, every pair once

        def hits(d):
            return any(m[s:s + K1] in kset for m in (F.store.mate1(int(d)), F.store.mate2(int(d))) for s in starts)
        for d in rng.choice(idx, size=min(2000, len(idx)), replace=False):
            assert hits(d)
        routed = set(idx.tolist())
        n_in = 0
        for d in rng.integers(0, 2 * N_PAIRS, size=20000):
            assert hits(d) == (int(d) in routed)
            n_in += int(d) in routed
    total = sum(len(v) for v in P["routes"].values())
    print("config %s: %d of the %d strand-doubled pairs are routed" % (F.cfg, total, 2 * N_PAIRS))
    assert total >= F.c["routed"] * 2 * N_PAIRS                           # (measured at configs[2]: 67 M of the 100 M strand-doubled pairs are routed)


def test_planted_transcripts_come_back(full):
    """The batch was sampled from 20,000 known transcripts (0.5 % substitutions per base): what the run reports is made of their
    k1-mers (precision), covers most of them (recall), and a good part of them comes back base for base, on either strand."""
    from shannon_amd import synth, _lib
    F = full
    K1 = F.K1
    iso, _ = synth.make_transcriptome(F.c["genes"], SEED, exon_len=F.c["exon_len"])
    A = np.frombuffer(b"ACGT", np.uint8)
    truth = [A[t].tobytes().decode() for t in iso]
    out = sorted(set(F.R.final.values()))
    # (precision and recall over the k1-mers whose hash falls into one class of eight, the same class on both sides: an unbiased
    # eighth of both sets -- the sorts and searches over all 2 x 10^8 k1-mers of the 4s transcriptome took this test four minutes)
    def eighth(keys):
        return keys[((keys * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(61)) == 0]
    tk = np.unique(eighth(canon_keys(_lib.string_windows(truth, K1)[0], K1)))
    ok = eighth(canon_keys(_lib.string_windows(out, K1)[0], K1))
    pos = np.minimum(np.searchsorted(tk, ok), len(tk) - 1)
    precision = float((tk[pos] == ok).mean())
    ou = np.unique(ok)
    pos = np.minimum(np.searchsorted(ou, tk), len(ou) - 1)
    recall = float((ou[pos] == tk).mean())
    outs = set(out)
    exact = sum(1 for t in truth if t in outs or rc(t) in outs)
    print("config %s planted transcripts: precision %.4f recall %.4f exact %d of %d, %d reported" % (F.cfg, precision, recall, exact, len(truth), len(out)))
    # measured at configs[2]: precision 0.9956, recall 0.9518, 5,185 of the 69,488 isoforms base for base, 40.6 k transcripts reported
    assert precision >= F.c["precision"] and recall >= F.c["recall"] and exact >= F.c["exact"] * len(truth)


def test_a_second_run_gives_the_same_bytes(full):
    """the whole output twice: same contigs, same partitions, same transcripts, same abundances -- and the digest bench.py reports"""
    import bench
    F = full
    R2 = F.run()
    assert R2.extension.contigs == F.R.extension.contigs
    assert list(R2.partitions) == list(F.R.partitions)
    for nm in F.R.partitions:
        assert R2.partitions[nm]["reconstructed_fasta"] == F.R.partitions[nm]["reconstructed_fasta"]
    assert R2.final == F.R.final
    assert bench._final_sha(R2.final) == F.sha
    import json
    want = json.load(open(DIGESTS)).get(F.cfg)                            # the digest the committed bench line of this config prints
    print("config %s transcripts_sha256_16 %s (recorded: %s)" % (F.cfg, F.sha, want))
    assert want is None or F.sha == want


ORACLE_CAP = 100_000       # routed pairs: what oracle/mbgraph.py (pure Python, ~80 us per pair + ~20 us per k1-mer row) finishes in seconds


def _partition_inputs(F, nm):
    """exactly what multibridging.main gets for partition nm: its k1-mer rows in file order (the weights never reach the output,
    multibridging.py:169 / mbgraph.py:735-767) and its reads up to the cutoff of multibridging.py:26-30, as strings"""
    from shannon_amd import pipeline
    P, K1 = F.R.partitioning, F.K1
    rb = P["k1mer_bytes"][nm]
    rb = rb() if callable(rb) else rb
    rows = [(bytes(rb[i:i + K1]).decode(), 1) for i in range(0, len(rb), K1)]
    cutoff = 10 * pipeline.n_kmer_nodes(rows, F.K) + 1
    idx = P["routes"][nm][:cutoff]
    return rows, [[F.store.mate1(int(d)) for d in idx], [F.store.mate2(int(d)) for d in idx]]


def test_partitions_of_the_run_through_the_oracle(full):
    """The back half at full size against the oracle: partitions of THIS run picked by rule among those with at most ORACLE_CAP routed
    pairs -- the smallest, the median, the one that bridged the most X-nodes, the one with the most known paths, the one with the
    most mate paths -- each with exactly its k1-mer rows and capped reads through oracle.mbgraph.run_partition (multibridging.py:
    209-269) and oracle.sparse_flow (algorithm_SF.py:373-613): canonical graph equal, transcripts equal record by record,
    abundances to 1e-6."""
    import time
    from oracle import mbgraph as omb, sparse_flow as osf
    from shannon_amd import mbgraph
    from golden_util import approx_eq
    from test_e2e_gpu import cmp_fasta
    F = full
    P = F.R.partitioning
    small = sorted((nm for nm in P["routes"] if len(P["routes"][nm]) <= ORACLE_CAP), key=lambda nm: len(P["routes"][nm]))
    assert len(small) >= 5
    log = {nm: F.R.partitions[nm]["log"] for nm in small}
    picks = {}
    for why, nm in (("smallest", small[0]), ("median under the cap", small[len(small) // 2]),
                    ("most bridged X-nodes", max(small, key=lambda nm: (sum(log[nm]["bridged"]), len(P["routes"][nm])))),
                    ("most known paths", max(small, key=lambda nm: (log[nm]["known_paths"], len(P["routes"][nm])))),
                    ("most mate paths", max(small, key=lambda nm: (log[nm]["mate_paths"], len(P["routes"][nm]))))):
        picks.setdefault(nm, why)
    if len(picks) < 3:                                         # rules that coincide: fill up with the largest ones under the cap
        for nm in reversed(small):
            picks.setdefault(nm, "largest under the cap")
            if len(picks) >= 3:
                break
    for nm, why in picks.items():
        t0 = time.time()
        rows, reads = _partition_inputs(F, nm)
        g, singles, comps = omb.run_partition(rows, reads, F.K, True)
        rec = F.R.partitions[nm]
        can_o, can_p = omb.canonical(singles, comps), mbgraph.canonical(rec["singles"], rec["components"])
        for k in can_o:
            assert approx_eq(can_p[k], can_o[k]), (nm, why, k)
        assert [l for l in g.log if "Bridged" in l] == ["Bridged %d nodes" % b for b in rec["log"]["bridged"]]
        sname = "bench_%s" % nm
        txt = ""
        for c, comp in enumerate(comps):
            tr = osf.sparse_flow_component(comp["nodes"], comp["edges"], comp["paths"], seed=1, comp_id=c)
            txt += osf.fasta_records(sname, str(c), tr)
        txt += osf.single_nodes_fasta(sname, singles)
        cmp_fasta(rec["reconstructed_fasta"], txt)
        print("config %s partition %s (%s): %d k1-mer rows, %d pairs, %d components, %d single nodes, %d transcripts == oracle; log %s (%.1f s)"
              % (F.cfg, nm, why, len(rows), len(reads[0]), len(comps), len(singles), txt.count(">"), [l for l in g.log if "Bridged" in l or "paths" in l], time.time() - t0))


def test_the_largest_partition_through_the_oracle(full):
    """The partition with the most routed pairs (3.9 M at configs[2]: the one that sets the wall time of the graph stage) through the
    ORACLE: oracle/mbgraph_fast.py = oracle/mbgraph.py with its two loops over all reads in numpy (tests/test_oracle_fast.py holds it
    against the sequential form), then oracle.sparse_flow -- canonical graph, "Bridged" log lines, transcripts and abundances
    against the run's record of that partition.  (Until round 4 only partitions of at most 100,000 pairs met the oracle.)"""
    import time
    from oracle import mbgraph as omb, mbgraph_fast as fast, sparse_flow as osf
    from shannon_amd import mbgraph, pipeline
    from golden_util import approx_eq
    from test_e2e_gpu import cmp_fasta
    F = full
    P, K1 = F.R.partitioning, F.K1
    # (the largest partition of all at configs[2]; at 4s the largest has 15.6 M pairs -- 223 s through the oracle, it passed in round 5 --
    # and the suite takes the largest one under a million pairs: the >= 1 M-pair class of the review is met by configs[2]'s 3.9 M)
    cap = 4_000_000 if F.cfg == "2" else 1_000_000
    nm = max((x for x in P["routes"] if len(P["routes"][x]) <= cap), key=lambda x: len(P["routes"][x]))
    t0 = time.time()
    rb = P["k1mer_bytes"][nm]
    rb = rb() if callable(rb) else rb
    rows = [(bytes(rb[i:i + K1]).decode(), 1) for i in range(0, len(rb), K1)]
    cutoff = 10 * pipeline.n_kmer_nodes(rows, F.K) + 1
    idx = np.asarray(P["routes"][nm][:cutoff], dtype=np.int64)
    b1, o1 = F.store.gather(idx, 1)
    b2, o2 = F.store.gather(idx, 2)
    L = int(o1[1] - o1[0])
    A, B = np.asarray(b1).reshape(len(idx), L), np.asarray(b2).reshape(len(idx), L)
    g, singles, comps = fast.run_partition_rows(rows, A, B, F.K)
    t1 = time.time()
    rec = F.R.partitions[nm]
    can_o, can_p = omb.canonical(singles, comps), mbgraph.canonical(rec["singles"], rec["components"])
    for k in can_o:
        assert approx_eq(can_p[k], can_o[k]), (nm, k)
    assert [l for l in g.log if "Bridged" in l] == ["Bridged %d nodes" % b for b in rec["log"]["bridged"]]
    sname = "bench_%s" % nm
    txt = ""
    for c, comp in enumerate(comps):
        tr = osf.sparse_flow_component(comp["nodes"], comp["edges"], comp["paths"], seed=1, comp_id=c)
        txt += osf.fasta_records(sname, str(c), tr)
    txt += osf.single_nodes_fasta(sname, singles)
    cmp_fasta(rec["reconstructed_fasta"], txt)
    print("config %s largest partition %s: %d k1-mer rows, %d pairs, %d distinct reads, %d components, %d transcripts == oracle; graph %.0f s, sparse flow %.0f s"
          % (F.cfg, nm, len(rows), len(idx), len(g.reads), len(comps), txt.count(">"), t1 - t0, time.time() - t1))


def test_a_median_partition_on_the_host_only_path(full):
    """the partition of median size (400 k-1 M routed pairs: beyond the Python oracle) through the native graph stage WITHOUT the
    device (shn_mbgraph_run, ctx = NULL: K-mer graph built and condensed sequentially from the k1-mer rows, reads as gathered text,
    seeds matched on the host) against the run's own record of it (unitigs on the GPU, distinct reads and seed scans on the device,
    reads named by rows): the same canonical graph.  Two code paths of the product, not an oracle -- but independent ones."""
    from shannon_amd import mbgraph, mbgraph_native, kmers_for_component as kfc
    from golden_util import approx_eq
    F = full
    P = F.R.partitioning
    names = sorted(P["routes"], key=lambda nm: len(P["routes"][nm]))
    nm = names[len(names) // 2]
    rows = kfc._rows_bytes(P["new_components"][nm], F.K1)
    idx = P["routes"][nm]
    b1, o1, rc1, enc = F.store.gather_codes(idx, 1)
    g = mbgraph_native.run_partition_handle(rows, len(rows) // F.K1, F.K, b1, o1, b1, o1, ctx=None, enc=enc, rc1=rc1, rc2=(1 - rc1).astype(np.uint8))
    singles, comps, _log = g.tables()
    g.close()
    rec = F.R.partitions[nm]
    a, b = mbgraph.canonical(singles, comps), mbgraph.canonical(rec["singles"], rec["components"])
    for k in a:
        assert approx_eq(a[k], b[k]), (nm, k)
    print("config %s partition %s: %d pairs, host-only graph == the run's (%d nodes)" % (F.cfg, nm, len(idx), len(a["nodes"])))
