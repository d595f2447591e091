"""The C restatements of the oracle (oracle/count_c.c, oracle/ext_c.c: counting and the greedy extension, the two stages that
dominate the reference's run time) against the Python restatement and the reference's own artefacts (tests/golden): the same k1-mer
table, the same walks in the same order, the same accepted contigs.  They are what lets the oracle check the HIP path at sizes the
pure-Python loops cannot reach (tests/test_midsize_gpu.py) and what bench.py times as the native CPU baseline."""
import numpy as np
import pytest
from golden_util import *
from oracle import seqs, count, extension, build_c

CASES = sorted(MANIFEST)


def canonical_table(tab, k1):
    """{k1-mer string: count} of the strand-doubled input -> (canonical keys ascending, counts): what oracle_count_canonical returns"""
    keys, cnts = [], []
    for s, c in tab.items():
        k = count.str_to_key(s)
        r = count.rc_key(k, k1)
        if k < r:
            keys.append(k); cnts.append(c)
        elif k == r:
            keys.append(k); cnts.append(c // 2)          # a palindrome is counted on both strands of the doubled input
    o = np.argsort(np.array(keys, dtype=np.uint64))
    return np.array(keys, dtype=np.uint64)[o], np.array(cnts, dtype=np.uint32)[o]


@pytest.mark.parametrize("name", DS_CASES)          # (the C restatement takes the canonical table of a strand-doubled input)
def test_c_extension_equals_the_python_oracle_and_the_reference(name):
    g = load_case(name)
    K, paired = g["K"], g["paired"]
    inp = load_inputs(name)
    dbl = read_files(name, inp)
    tab = count.count_k1mers_dict([r for f in dbl for r in f], K + 1)
    items = [(k, tab[k]) for k in sorted(tab, reverse=True)]
    kmers, k1 = extension.load_kmers(items)
    want = list(extension.python_walks(kmers, k1, 3))
    keys, cnts = canonical_table(tab, K + 1)
    got = build_c.extend(keys, cnts, K + 1, 3)
    assert len(got) == len(want)
    assert got == [(c, int(w), n) for c, w, n in want]           # the same walks, in the same (seed) order
    psize = MANIFEST[name].get("partition_size", 500)
    res = extension.run_correction(items, comp_size_threshold=psize, walks=got)
    assert res.contigs == g["contigs"]                           # ... and the reference's accepted contigs


def test_c_counter_feeds_the_c_extension():
    """codes -> count_c -> ext_c on a synthetic paired input == the Python oracle on the same reads"""
    from shannon_amd import synth
    (r1, r2), _ = synth.make_dataset(3000, 3, seed=17)
    codes = np.concatenate([r1, r2])
    keys, cnts, nw = build_c.count_canonical(codes, 26, True)
    A = np.frombuffer(b"ACGT", np.uint8)
    s1 = [A[r].tobytes().decode() for r in r1]
    s2 = [A[r].tobytes().decode() for r in r2]
    dbl = list(seqs.double_strand_paired(s1, s2))
    tab = count.count_k1mers_dict([r for f in dbl for r in f], 26)
    k2, c2 = canonical_table(tab, 26)
    assert np.array_equal(keys, k2) and np.array_equal(cnts, c2)
    kmers, k1 = extension.load_kmers([(k, tab[k]) for k in sorted(tab, reverse=True)])
    want = [(c, int(w), n) for c, w, n in extension.python_walks(kmers, k1, 3)]
    assert build_c.extend(keys, cnts, 26, 3) == want


def test_oracle_front_over_c_walks_equals_the_python_oracle():
    """the chained oracle that tests/test_midsize_gpu.py runs at 10^6 reads (C count + C walks, then run_correction(walks=...) over the
    dictionary restricted to the k1-mers of acceptable walks) against the plain Python oracle pipeline on a small batch: same
    contigs, allowed dictionary, partitions, routed reads and k1-mer files"""
    from shannon_amd import synth
    from oracle import partition
    from test_midsize_gpu import oracle_front
    iso, _ = synth.make_transcriptome(12, 5)
    r1, r2 = synth.sample_pairs(iso, 6000, 5)
    K = 25
    ok, walks, res, pv, nc, o1, o2, files = oracle_front(r1, r2, K)
    A = np.frombuffer(b"ACGT", np.uint8)
    s1 = [A[r].tobytes().decode() for r in r1]
    s2 = [A[r].tobytes().decode() for r in r2]
    dbl = list(seqs.double_strand_paired(s1, s2))
    tab = count.count_k1mers_dict([r for f in dbl for r in f], K + 1)
    want = extension.run_correction([(k, tab[k]) for k in sorted(tab, reverse=True)])
    assert res.contigs == want.contigs and len(want.contigs) > 20
    assert res.allowed == want.allowed
    assert res.single_contigs == want.single_contigs and res.remaining == want.remaining
    nc2, k2c2 = partition.build_partitions([], [], None, want.remaining, want.allowed, K)
    assert list(nc) == list(nc2) and all(nc[n] == nc2[n] for n in nc)
    w1, w2 = partition.route_reads_paired(dbl[0], dbl[1], nc2, k2c2, K)
    assert o1 == w1 and o2 == w2
    assert files == partition.partition_k1mers(nc2, k2c2, K)[0]


def test_vectorised_routing_equals_the_python_routing():
    """oracle.partition.route_pairs_matrix (numpy over code matrices: what bench.py's CPU baseline routes a million reads with)
    against route_reads_paired over the strand-doubled strings: the same pairs in the same order, several partitions, k1-mers that
    lie in two partitions, reads with a base outside ACGT"""
    from shannon_amd import synth
    from oracle import partition
    from test_midsize_gpu import oracle_front
    iso, _ = synth.make_transcriptome(30, 5)
    r1, r2 = synth.sample_pairs(iso, 12000, 5)
    ok, walks, res, pv, nc, o1, o2, files = oracle_front(r1, r2, 25)
    r1 = r1.copy(); r2 = r2.copy()
    r1[5, 7] = 4; r2[9, 3] = 4; r1[100, 99] = 4
    A = np.frombuffer(b"ACGTN", np.uint8)
    d1, d2 = seqs.double_strand_paired([A[r].tobytes().decode() for r in r1], [A[r].tobytes().decode() for r in r2])
    # partitions of three contigs each, and every contig a second time in an "r2_" partition (k1-mers with two owners)
    contigs = [c for lst in res.remaining for c in lst]
    assert len(contigs) >= 12
    big = [contigs]
    parts = [[i // 3 for i in range(len(contigs))]]
    parts2 = [[(i // 2) % 4 for i in range(len(contigs))]]
    nc2, k2c2 = partition.build_partitions(big, parts, parts2, [], res.allowed, 25)
    w1, w2 = partition.route_reads_paired(d1, d2, nc2, k2c2, 25)
    got = partition.route_pairs_matrix(r1, r2, nc2, 25)
    assert len(nc2) >= 8 and sum(len(v) for v in got.values()) > 10000
    for c in nc2:
        idx = got[c].tolist()
        assert [d1[i] for i in idx] == w1[c] and [d2[i] for i in idx] == w2[c], c
