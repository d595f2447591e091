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


def test_partitions_are_dealt_by_load():
    from shannon_amd import distributed
    o = distributed.deal_partitions([5, 100, 7, 7, 60, 1], 3)
    acc = np.bincount(o, weights=[5, 100, 7, 7, 60, 1], minlength=3)
    assert o[1] == 0 and o[4] == 1 and acc.max() == 100 and acc.min() >= 19
    assert distributed.deal_partitions([], 4).tolist() == [] and distributed.deal_partitions([3, 3, 3], 1).tolist() == [0, 0, 0]


@pytest.mark.parametrize("name,port,world,chunk", [("syn_pe_s0", 29611, 2, None), ("syn_se_s5", 29612, 2, None), ("syn_part_s33", 29613, 2, "997"),
                                                   ("syn_part_s33", 29614, 3, None), ("syn_pe_s0", 29615, 3, "500"),
                                                   ("syn_pe_ss_s69", 29617, 2, None), ("syn_se_ss_s53", 29618, 3, None)])     # -s / --strand_specific
def test_ranks_equal_single_process(name, port, world, chunk, tmp_path):
    """world_size 2 and 3 (gloo); chunk: every variable-size collective in rounds of that many elements (exchange.chunk_elems)"""
    from oracle import pipeline as opipe
    out = str(tmp_path / "res.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if chunk:
        env["SHN_COLL_CHUNK"] = chunk
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), name, out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    got = json.load(open(out))
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
