"""CPU: oracle/mbgraph_fast.py (the oracle's two loops over all reads in numpy) against oracle/mbgraph.py itself, whole partitions:
the same Read objects (insertion numbers, copy counts, mates), the same graphs, the same log lines."""
import numpy as np
import pytest
from golden_util import *


def _partitions(n_pairs, n_genes, seed, paired=True, K=25):
    """(k1-mer rows, reads1, reads2) of every partition of a synthetic run, made by the oracle's own front half"""
    from shannon_amd import synth
    from oracle import seqs, count, extension, partition
    iso, _ = synth.make_transcriptome(n_genes, seed, n_isoforms=(3, 6), n_exons=(6, 12))       # (isoform-rich genes: multi-contig components)
    r1, r2 = synth.sample_pairs(iso, n_pairs, seed)
    A = np.frombuffer(b"ACGT", np.uint8)
    s1, s2 = [A[r].tobytes().decode() for r in r1], [A[r].tobytes().decode() for r in r2]
    if not paired:
        s1, s2 = s1 + s2, None
    dbl = list(seqs.double_strand_paired(s1, s2)) if paired else [seqs.double_strand_single(s1)]
    tab = count.count_k1mers_dict([r for f in dbl for r in f], K + 1)
    res = extension.run_correction([(k, tab[k]) for k in sorted(tab, reverse=True)])
    nc, k2c = partition.build_partitions([], [], None, res.remaining, res.allowed, K)
    if paired:
        o1, o2 = partition.route_reads_paired(dbl[0], dbl[1], nc, k2c, K)
    else:
        o1, o2 = partition.route_reads(dbl[0], nc, k2c, K), None
    files, _ = partition.partition_k1mers(nc, k2c, K)
    return [(files[nm], o1[nm], o2[nm] if paired else None) for nm in nc]


def _rows(strings):
    return np.frombuffer("".join(strings).encode(), dtype=np.uint8).reshape(len(strings), -1) if strings else np.zeros((0, 1), np.uint8)


@pytest.mark.parametrize("paired,n_genes,seed", [(True, 3, 11), (False, 2, 5), (True, 8, 4)])
def test_vectorised_loops_equal_the_sequential_oracle(paired, n_genes, seed):
    from oracle import mbgraph as omb, mbgraph_fast as fast
    n_checked = 0
    for rows, q1, q2 in _partitions(12000, n_genes, seed, paired):
        if not q1:
            continue
        g0, s0, c0 = omb.run_partition(rows, [q1, q2] if paired else [q1], 25, paired)
        g1, s1, c1 = fast.run_partition_rows(rows, _rows(q1), _rows(q2) if paired else None, 25)
        # the loader: the same reads in the same order with the same counts, mates and flags
        a, b = list(g0.reads.values()), list(g1.reads.values())
        assert [(r.rid, r.bases, r.copy_count, r.mate_pair, r.mate.rid if r.mate else None) for r in a] == \
               [(r.rid, r.bases, r.copy_count, r.mate_pair, r.mate.rid if r.mate else None) for r in b]
        assert g0.log == g1.log
        assert omb.canonical(s0, c0) == omb.canonical(s1, c1)
        n_checked += 1
    assert n_checked >= 1


def test_duplicates_mates_and_lower_case():
    """a read that occurs as both mates, pairs repeated in different roles, lower-case input: what the LAST occurrence leaves"""
    from oracle import mbgraph as omb, mbgraph_fast as fast
    rng = np.random.default_rng(3)
    base = ["".join("ACGT"[c] for c in rng.integers(0, 4, 60)) for _ in range(6)]
    q1 = [base[0], base[1], base[0], base[2], base[3].lower(), base[1], base[4], base[4]]
    q2 = [base[1], base[0], base[2], base[2], base[3], base[5], base[4], base[0]]
    g0, g1 = omb.MBGraph(25, 60), fast.FastMBGraph(25, 60)
    for g in (g0, g1):
        g.nodes = [None] * 10                       # (the read cap of multibridging.py:26-30 counts nodes)
    g0.load_mated_reads(q1, q2)
    g1.load_mated_rows(_rows(q1), _rows(q2))
    key = lambda g: [(r.rid, r.bases, r.copy_count, r.mate_pair, r.mate.rid) for r in g.reads.values()]
    assert key(g0) == key(g1)
    g0, g1 = omb.MBGraph(25, 60), fast.FastMBGraph(25, 60)
    g0.load_reads(q1 + q2)                            # no nodes: the cap is read 0 alone
    g1.load_read_rows(_rows(q1 + q2))
    assert [(r.rid, r.bases, r.copy_count) for r in g0.reads.values()] == [(r.rid, r.bases, r.copy_count) for r in g1.reads.values()]
