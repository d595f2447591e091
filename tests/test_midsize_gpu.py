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


def oracle_front(r1, r2, K):
    """the chained oracle up to the partitions' inputs: C count + C walks (oracle/count_c.c, oracle/ext_c.c), then the oracle's
    Python from the walks on: accept filter, duplicate_check, contig graph, components, partitions, read routing, k1-mer files"""
    from oracle import build_c, extension, partition, seqs
    from golden_util import part_vectors
    k1 = K + 1
    ok, oc, nw = build_c.count_canonical(np.concatenate([r1, r2]), k1, True)
    walks = build_c.extend(ok, oc, k1, 3)
    # the dictionary run_correction reads when the walks are given: the weights of the k1-mers of walks that can be accepted
    A = np.frombuffer(b"ACGT", np.uint8)
    code = np.zeros(256, np.uint64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    items = {}
    pw = (np.uint64(4) ** np.arange(k1 - 1, -1, -1, dtype=np.uint64)).astype(np.uint64)
    for contig, _w, _n in walks:
        if len(contig) < 75:
            continue
        c = code[np.frombuffer(contig.encode(), np.uint8)]
        fw = np.lib.stride_tricks.sliding_window_view(c, k1) @ pw
        rv = np.lib.stride_tricks.sliding_window_view(np.uint64(3) - c, k1) @ pw[::-1]
        cnt = oc[np.searchsorted(ok, np.minimum(fw, rv))].tolist()
        for i, w in enumerate(cnt):
            items[contig[i:i + k1]] = w
    res = extension.run_correction(sorted(items.items(), reverse=True), walks=walks)
    pv = [part_vectors(len(b[0]), 500) for b in res.big_components] or None
    nc, k2c = partition.build_partitions([b[0] for b in res.big_components], [p[0] for p in pv] if pv else [], [p[1] for p in pv] if pv else None,
                                         res.remaining, res.allowed, K)
    s1 = [A[r].tobytes().decode() for r in r1]
    s2 = [A[r].tobytes().decode() for r in r2]
    d1, d2 = seqs.double_strand_paired(s1, s2)
    o1, o2 = partition.route_reads_paired(d1, d2, nc, k2c, K)
    files, _ = partition.partition_k1mers(nc, k2c, K)
    return ok, walks, res, pv, nc, o1, o2, files


def test_whole_path_equals_the_oracle_at_a_million_reads():
    """1 M reads (500 k pairs) of 500 genes, K = 25, through the WHOLE path on the device and through the chained oracle: the front
    through the C restatements (oracle/count_c.c, oracle/ext_c.c: the table and every walk), then the oracle's Python from the walks
    on -- accept filter, duplicate_check, contig graph, components (extension.run_correction(walks=...)), partitions, read routing,
    multibridged graph (mbgraph.run_partition), sparse flow, final merge.  Contigs, partitions, every partition's canonical graph,
    transcripts record by record (abundances to 1e-6) and the final file equal.  (The goldens reach 28 k pairs, the full-size runs are
    checked through properties and a few partitions: this is the one run of 10^6 reads compared stage by stage.)"""
    import time
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from shannon_amd import device, pipeline, mbgraph
    from oracle import build_c, extension, partition, mbgraph as omb, sparse_flow as osf, post as opost, seqs
    from golden_util import approx_eq, part_vectors
    from test_e2e_gpu import cmp_fasta
    K, k1 = 25, 26
    r1, r2 = bench.gen_reads(500_000, 20240501, 500, torch.device("cuda", 0), read_seed=7)
    t0 = time.time()
    ok, walks, res, pv, nc, o1, o2, files = oracle_front(r1, r2, K)
    t_front = time.time() - t0
    ctx = device.Context(0)
    R = pipeline.assemble(ctx, r1, r2, K=K, sample="mid", seed=1, part_vectors=pv)
    assert R.extension.contigs == res.contigs and len(res.contigs) > 500
    assert list(R.partitions) == list(nc)
    lines = [">Single_%d\n%s\n" % (i, c) for i, c in enumerate(res.single_contigs)]
    lines = "".join(lines).splitlines(True)
    n_routed = 0
    for name in nc:
        assert R.partitions[name]["n_reads_routed"] == len(o1[name])
        n_routed += len(o1[name])
        g, singles, comps = omb.run_partition(files[name], [o1[name], o2[name]], K, True)
        rec = R.partitions[name]
        a, b = omb.canonical(singles, comps), mbgraph.canonical(rec["singles"], rec["components"])
        for k in a:
            assert approx_eq(b[k], a[k]), (name, k)
        sname = "mid_%s" % name
        txt = ""
        for c, comp in enumerate(comps):
            txt += osf.fasta_records(sname, str(c), osf.sparse_flow_component(comp["nodes"], comp["edges"], comp["paths"], seed=1, comp_id=c))
        txt += osf.single_nodes_fasta(sname, singles)
        cmp_fasta(rec["reconstructed_fasta"], txt)
        lines += txt.splitlines(True)
    assert R.final == opost.finalize(lines, True)
    print("1 M reads: %d k1-mers, %d walks, %d contigs, %d partitions, %d routed pairs, %d final transcripts == oracle (oracle front %.0f s, all %.0f s)"
          % (len(ok), len(walks), len(res.contigs), len(nc), n_routed, len(R.final), t_front, time.time() - t0))
    ctx.close()
