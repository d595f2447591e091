"""CPU, world_size 2 and 3 (gloo): the multi-GPU choreography of shannon_amd.distributed -- bucket
exchange, table replication, global read order / caps, partition ownership, FASTA gather --
reproduces the single-process result on the concatenated reads."""
import json, os, subprocess, sys
import numpy as np
import pytest
from golden_util import *
from conftest import ROOT


def test_owner_hash_matches_device_constant():
    from shannon_amd import exchange
    src = open(os.path.join(ROOT, "shannon_amd", "csrc", "count.hip")).read()
    assert "#define SHARD_SALT 0x%XULL" % exchange.SHARD_SALT in src
    o = exchange.owner_of(np.arange(1000, dtype=np.uint64), 8)
    assert o.min() == 0 and o.max() == 7 and np.bincount(o).min() > 80


def test_minimizer_owner_keeps_neighbours_together():
    """exchange.owner_of_minimizer (host mirror of shn_owner_minimizer, csrc/common.h): the constants are the device's, the ranks are
    balanced, a k-mer and its reverse complement have one owner, and a k-mer's successor mostly stays with it -- what keeps the
    edges of the k1-mer graph inside a shard"""
    from shannon_amd import exchange
    src = open(os.path.join(ROOT, "shannon_amd", "csrc", "common.h")).read()
    assert "#define SHN_OWNER_M %d" % exchange.OWNER_M in src and "0x%XULL" % exchange.OWNER_SALT in src
    rng = np.random.default_rng(7)
    k, W = 26, 8
    keys = rng.integers(0, 1 << 52, 20000, dtype=np.uint64)
    own = exchange.owner_of_minimizer(keys, k, True, W)
    assert own.min() == 0 and own.max() == W - 1 and np.bincount(own, minlength=W).min() > 20000 // W // 2
    rc = np.zeros_like(keys); x = keys.copy()
    for _ in range(k):
        rc = (rc << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3))); x >>= np.uint64(2)
    assert np.array_equal(exchange.owner_of_minimizer(rc, k, True, W), own)                    # strand-independent when canonical
    succ = ((keys << np.uint64(2)) | rng.integers(0, 4, len(keys), dtype=np.uint64)) & np.uint64((1 << 52) - 1)
    same = (exchange.owner_of_minimizer(succ, k, True, W) == own).mean()
    assert same > 0.8, same                                                                     # (7 of 8 share the minimizer, the rest 1 in W)
    assert (exchange.owner_of_minimizer(keys, k, False, W) != own).any()                        # the plain (strand-specific) rule is another one


def test_partitions_are_dealt_by_load():
    from shannon_amd import distributed
    o = distributed.deal_partitions([5, 100, 7, 7, 60, 1], 3)
    acc = np.bincount(o, weights=[5, 100, 7, 7, 60, 1], minlength=3)
    assert o[1] == 0 and o[4] == 1 and acc.max() == 100 and acc.min() >= 19
    assert distributed.deal_partitions([], 4).tolist() == [] and distributed.deal_partitions([3, 3, 3], 1).tolist() == [0, 0, 0]


def test_owner_shard_labelling_in_numpy_gives_the_components():
    """tests/cc_reference.py: NumpyComponents (the CPU stand-in for shn_cc_* behind distributed.component_table) with 3 played ranks
    against the components of the whole set -- the restatement the gloo tests below rely on (the device kernels are held against
    the same reference in tests/test_cc_shards_gpu.py)"""
    from cc_reference import NumpyComponents, reference_labels, same_partition
    from shannon_amd import exchange, synth
    from oracle import count, seqs
    (q1, q2), _ = synth.make_dataset(1500, 4, seed=3)
    reads = ["".join("ACGT"[c] for c in r) for r in np.concatenate([q1, q2])]
    reads += [seqs.reverse_complement(r) for r in reads]
    k1, W = 26, 3
    keys, cnts = count.count_k1mers_packed(reads, k1)
    canon = np.array([int(k) <= count.rc_key(k, k1) for k in keys], dtype=bool)
    keys, cnts = keys[canon], cnts[canon]
    own = exchange.owner_of_minimizer(keys, k1, True, W)
    ccs = [NumpyComponents(keys[own == r], cnts[own == r], W, r, k1, True) for r in range(W)]
    base = [sum(c.n for c in ccs[:r]) for r in range(W)]
    sent = [c.queries() for c in ccs]
    edges = []
    for d in range(W):
        offs = [np.concatenate([[0], np.cumsum(s[2])]) for s in sent]
        import torch
        rk = torch.cat([sent[s][0][offs[s][d]:offs[s][d + 1]] for s in range(W)])
        rl = torch.cat([sent[s][1][offs[s][d]:offs[s][d + 1]] for s in range(W)])
        edges.append(ccs[d].answer(rk, rl, [int(sent[s][2][d]) for s in range(W)], base))
    ge = torch.cat(edges)
    assert ge.numel() > 0
    ids, labels = ccs[0].solve(ge, sum(c.n for c in ccs) + 1)
    gl = np.concatenate([ccs[r].labels(base[r], ids, labels).numpy() for r in range(W)])
    allk = np.concatenate([c.keys for c in ccs])
    ref = reference_labels(allk, k1, True)
    assert len(np.unique(ref)) > 3 and same_partition(gl, ref)


@pytest.mark.parametrize("name,port,world", [("syn_pe_s0", 29651, 2), ("syn_part_s33", 29652, 3), ("syn_se_ss_s53", 29653, 2)])
def test_ranks_with_owner_shard_labelling_equal_single_process(name, port, world, tmp_path):
    """the default N-rank path (no replicated table: minimizer-sharded owners, components labelled on the shards, whole components
    to their rank -- distributed.component_table) over gloo with the numpy stand-in for the device kernels; the worker checks that
    every component ends on one rank and that the exchanged tables are the job's k1-mers"""
    _ranks_equal_single(name, port, world, None, tmp_path, {"SHN_TEST_OWNER_LABELS": "1"})


@pytest.mark.parametrize("name,port,world,chunk", [("syn_pe_s0", 29611, 2, None), ("syn_se_s5", 29612, 2, None), ("syn_part_s33", 29613, 2, "997"),
                                                   ("syn_part_s33", 29614, 3, None), ("syn_pe_s0", 29615, 3, "500"),
                                                   ("syn_pe_ss_s69", 29617, 2, None), ("syn_se_ss_s53", 29618, 3, None)])     # -s / --strand_specific
def test_ranks_equal_single_process(name, port, world, chunk, tmp_path):
    """world_size 2 and 3 (gloo); chunk: every variable-size collective in rounds of that many elements (exchange.chunk_elems)"""
    _ranks_equal_single(name, port, world, chunk, tmp_path, {})


def _ranks_equal_single(name, port, world, chunk, tmp_path, env_extra):
    from oracle import pipeline as opipe
    out = str(tmp_path / "res.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    if chunk:
        env["SHN_COLL_CHUNK"] = chunk
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), name, out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    got = json.load(open(out))
    assert got["owner_labelling_ran"] == (env_extra.get("SHN_TEST_OWNER_LABELS") == "1")
    m = MANIFEST[name]
    g = load_case(name)
    inp = load_inputs(name)
    psize = m.get("partition_size", 500)
    pv = [part_vectors(len(b["contigs"]), psize) for b in g["big_components"]] or None
    ref = opipe.assemble(inp[0], inp[1] if m["paired"] else None, K=m["K"], partition_size=psize, sample="s", seed=m["sf_seed"], part_vectors=pv,
                         double_stranded=not m.get("strand_specific"))
    assert got["contigs"] == ref["contigs"]
    assert list(got["partitions"]) == list(ref["partitions"])
    for nm in ref["partitions"]:
        assert got["partitions"][nm] == ref["partitions"][nm]["reconstructed_fasta"]
    assert got["final"] == ref["final"]


def test_a_failing_graph_stage_is_told_to_every_rank(tmp_path):
    """rank 1's graph stage raises: the error travels with the FASTA gather, all ranks raise the same message, none hangs"""
    out = str(tmp_path / "res.json")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29616", os.path.join(ROOT, "tests", "dist_worker.py"), "syn_se_s5", out, "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    for r in (0, 1):
        msg = open(out + ".rank%d" % r).read()
        assert "graph stage failed on rank 1" in msg and "boom on purpose" in msg, msg


def test_read_pieces_travel_as_bytes():
    """The capped read sets of the partitions go to their owners as one byte buffer per destination (exchange.pack_read_pieces
    / unpack_read_pieces around all_to_all_bytes): round trip, empty pieces and empty buffers included."""
    from shannon_amd import exchange
    rng = np.random.default_rng(5)
    items = []
    for p, n in ((3, 5), (0, 0), (9, 1000)):
        items.append((p, np.sort(rng.integers(0, 1 << 40, n)).astype(np.int64),
                      (rng.integers(0, 4, (n, 100), dtype=np.uint8), rng.integers(0, 2, n, dtype=np.uint8))))
    buf = exchange.pack_read_pieces(items)
    assert buf.dtype == np.uint8 and buf.ndim == 1
    back = exchange.unpack_read_pieces(buf)
    assert [b[0] for b in back] == [3, 0, 9]
    for (p, g, (rows, rc)), (p2, g2, (rows2, rc2)) in zip(items, back):
        assert np.array_equal(g, g2) and np.array_equal(rows, rows2) and np.array_equal(rc, rc2) and rows2.shape == rows.shape
    assert exchange.unpack_read_pieces(exchange.pack_read_pieces([])) == []
    assert exchange.unpack_read_pieces(np.zeros(0, np.uint8)) == []


def test_strand_specific_is_refused_by_ops_that_do_not_implement_it():
    """-s / --ss on the N-rank path needs ops that count forward, route plain indices and pair reads_1 with RC(reads_2): ops that
    do not say they do are refused instead of run double-stranded"""
    from shannon_amd import distributed
    with pytest.raises(NotImplementedError, match="strand_specific"):
        distributed.assemble_distributed(object(), double_stranded=False)
