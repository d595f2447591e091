"""Parity at a size between the goldens (thousands of reads) and BASELINE's configs (10^7-10^8 reads, properties only): 4 M reads
of 2,000 genes, K = 25 and K = 31, through the HIP counting and extension kernels and through the oracle's C restatements
(oracle/count_c.c, oracle/ext_c.c -- pinned against the Python restatement and the reference's artefacts in tests/test_oracle_c.py):
the same k1-mer table and EVERY walk the same, in seed order -- contig strings, weights, lengths (~10^6 walks, the bulk rounds of
the walker and the chunked / partition-pipeline paths of the counter included), and the accepted contigs after the accept filter,
duplicate_check and the contig graph through the oracle's Python over those walks."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [25, 31])
def test_every_walk_equals_the_c_oracle(K):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from shannon_amd import device, extension_correction as ec
    from oracle import build_c
    k1 = K + 1
    dev = torch.device("cuda", 0)
    r1, r2 = bench.gen_reads(2_000_000, 20240501, 2000, dev, read_seed=99, exon_len=(80, 600) if K == 25 else (80, 3000))
    ctx = device.Context(0)
    d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
    t = device.count_k1mers(ctx, [d1, d2], k1, both_strands=True)
    keys, cnts = t.download()
    o = np.argsort(keys, kind="stable")
    ok, oc, nw = build_c.count_canonical(np.concatenate([r1, r2]), k1, True)
    assert t.total == nw and np.array_equal(keys[o], ok) and np.array_equal(cnts[o].astype(np.uint64), oc.astype(np.uint64))
    want = build_c.extend(ok, oc, k1, 3)
    ext = ec.Extension(ctx, t, 3)
    rank, nr, nl, tw = ext.live_stats()
    assert len(rank) == len(want) > 50_000
    assert tw.tolist() == [w for _, w, _ in want]
    assert (nr.astype(np.int64) + nl.astype(np.int64) + 1).tolist() == [n for _, _, n in want]
    got = ext.emit(rank, k1 + nr.astype(np.int64) + nl.astype(np.int64))
    assert got == [c for c, _, _ in want]
    print("K=%d: %d k1-mers, %d walks (%d seeds, %d rounds, %d steps) equal the C oracle" % (K, len(ok), len(want), ext.n_walks, ext.iterations, ext.total_steps))
    ext.close(); t.close(); d1.close(); d2.close(); ctx.close()
