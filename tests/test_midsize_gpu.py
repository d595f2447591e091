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


def test_gpu_contig_stage_and_bulk_walker_equal_the_sequential_stage_at_10m_reads():
    """10 M reads of 2,000 genes (73 M k1-mers: above the 20 M at which the product switches to the GPU contig stage, bulk rounds of
    the thread walker on 23 M seeds): the contig stage both ways -- the reference's sequential loop in native host code beside the
    walks (SHN_CONTIG_GPU=0) and the fixpoint rounds on the device after them (SHN_CONTIG_GPU=1: what BASELINE configs[2] runs) -- each
    twice: contigs, contig connections and components identical in all four runs.  (tools/check_contig_paths.py as a test.)"""
    import hashlib
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from shannon_amd import device, extension_correction as ec
    r1, r2 = bench.gen_reads(5_000_000, 20240501, 2000, torch.device("cuda", 0))
    ctx = device.Context(0)
    sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
    table = device.count_k1mers(ctx, sets, 26, True)
    assert len(table) >= 20_000_000
    sigs = []
    old = os.environ.get("SHN_CONTIG_GPU")
    try:
        for mode in ("0", "1", "0", "1"):
            os.environ["SHN_CONTIG_GPU"] = mode
            res = ec.run_correction(ctx, table, 3, 75, 500, want_allowed=False)
            hc = hashlib.sha256(np.asarray(res.conn_off, np.int64).tobytes() + np.asarray(res.conn_nb, np.int64).tobytes() +
                                np.asarray(res.conn_w, np.int64).tobytes()).hexdigest()
            sigs.append((hashlib.sha256("\n".join(res.contigs).encode()).hexdigest(), hc,
                         hashlib.sha256(np.asarray(res.comp_members, np.int64).tobytes()).hexdigest(), len(res.contigs)))
    finally:
        if old is None:
            os.environ.pop("SHN_CONTIG_GPU", None)
        else:
            os.environ["SHN_CONTIG_GPU"] = old
    assert len(set(sigs)) == 1 and sigs[0][3] > 5000, sigs
    table.close()
    for x in sets:
        x.close()
    ctx.close()
