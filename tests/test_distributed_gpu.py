"""The multi-GPU code path (shannon_amd.distributed.assemble_distributed with GpuOps: HIP kernels through the C ABI,
collectives on the "nccl" = RCCL backend) run as a one-rank job on cuda:0 and compared with the single-process
pipeline on the same inputs.  (world_size 2 is covered on CPU with gloo in test_distributed_cpu.py.)"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group():
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("paired,n_genes,seed", [(True, 3, 11), (False, 2, 5), (True, 12, 4)])
def test_distributed_path_matches_single_process(group, paired, n_genes, seed):
    from shannon_amd import device, synth, pipeline, distributed, kmers_for_component as kfc
    (q1, q2), _ = synth.make_dataset(12000, n_genes, seed=seed)
    if not paired:
        q1, q2 = np.concatenate([q1, q2]), None
    ctx = device.Context(0)
    d1 = device.Reads.from_codes(ctx, q1)
    d2 = device.Reads.from_codes(ctx, q2) if paired else None
    store = kfc.ReadStore(q1, q2)
    try:
        ref = pipeline.assemble_resident(ctx, d1, d2, store, K=25, sample="t", seed=1)
        ops = distributed.GpuOps(ctx, d1, d2, store, 25)
        got = distributed.assemble_distributed(ops, 25, 500, "t", 1)
        assert got["contigs"] == ref.extension.contigs
        assert list(got["partitions"]) == list(ref.partitions)
        for name in ref.partitions:
            assert got["partitions"][name] == ref.partitions[name]["reconstructed_fasta"]
        assert got["final"] == ref.final
    finally:
        d1.close()
        if d2 is not None:
            d2.close()
        ctx.close()


def test_array_gather_and_slice_ingest_on_rccl_with_device_tensors(group, tmp_path):
    """the collectives round 6 added, on the "nccl" = RCCL backend with device tensors (one rank): exchange.all_gather_arrays (the
    candidates of the sharded contig stage as tensors) and distributed.ingest_rank_slice (record counts + sparse index gathered,
    the rank's stretch of the file through shn_reads_ingest) -- over gloo with four ranks in tests/test_ingest_ranks.py"""
    import torch
    from shannon_amd import exchange, distributed, device
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    mine = (rng.integers(0, 1 << 40, 1000).astype(np.int64), rng.integers(0, 1 << 60, 1000, dtype=np.uint64), np.arange(1001, dtype=np.uint64),
            rng.integers(0, 4, 123457).astype(np.uint8), np.zeros(0, np.uint8))
    got = exchange.all_gather_arrays(mine, exchange.coll_device(dev, None), None, "test arrays")
    assert len(got) == 1 and all(a.dtype == b.dtype and np.array_equal(a, b) for a, b in zip(got[0], mine))
    codes = rng.integers(0, 4, (5000, 100), dtype=np.uint8)
    path = str(tmp_path / "r.fasta")
    with open(path, "w") as f:
        for i, row in enumerate(codes):
            f.write(">r%d\n%s\n" % (i, np.frombuffer(b"ACGT", np.uint8)[row].tobytes().decode()))
    st = {}
    mats, n = distributed.ingest_rank_slice([path], 0, 1, None, exchange.coll_device(dev, None), stats=st)
    assert n == 5000 and np.array_equal(mats[0], codes) and st["bytes_scanned"] <= 2 * os.path.getsize(path) + 64


@pytest.mark.parametrize("paired,n_genes,seed,big", [(True, 12, 4, False), (False, 6, 5, True)])
def test_owner_shard_path_on_rccl_with_one_rank(group, paired, n_genes, seed, big, monkeypatch):
    """SHN_OWNER_LABELS=2 sends a one-rank job through the N-rank path (shard by minimizer, queries, edges, component exchange): every
    collective of it on the "nccl" = RCCL backend with device tensors -- what the shared-GPU tests below do over gloo"""
    from shannon_amd import device, synth, pipeline, distributed, kmers_for_component as kfc
    monkeypatch.setenv("SHN_OWNER_LABELS", "2")
    if big:
        monkeypatch.setenv("SHN_CONTIG_GPU", "1")
    (q1, q2), _ = synth.make_dataset(12000, n_genes, seed=seed)
    if not paired:
        q1, q2 = np.concatenate([q1, q2]), None
    ctx = device.Context(0)
    d1 = device.Reads.from_codes(ctx, q1)
    d2 = device.Reads.from_codes(ctx, q2) if paired else None
    store = kfc.ReadStore(q1, q2)
    try:
        ref = pipeline.assemble_resident(ctx, d1, d2, store, K=25, sample="t", seed=1)
        ops = distributed.GpuOps(ctx, d1, d2, store, 25)
        T = {}
        got = distributed.assemble_distributed(ops, 25, 500, "t", 1, timings=T)
        assert "x:component exchange" in T and ops.component_table_sizes[0] == ops.component_table_sizes[1] == ops.component_table_sizes[2]
        assert got["contigs"] == ref.extension.contigs
        assert list(got["partitions"]) == list(ref.partitions)
        for name in ref.partitions:
            assert got["partitions"][name] == ref.partitions[name]["reconstructed_fasta"]
        assert got["final"] == ref.final
    finally:
        d1.close()
        if d2 is not None:
            d2.close()
        ctx.close()


@pytest.mark.parametrize("world,paired,n_genes,seed,port,big,ss", [(2, True, 3, 11, 29621, False, False), (3, True, 12, 4, 29622, False, False),
                                                                   (2, False, 2, 5, 29623, False, False), (4, True, 40, 8, 29624, False, False),
                                                                   (2, True, 40, 8, 29625, True, False), (3, True, 12, 4, 29626, True, False),
                                                                   (4, False, 30, 6, 29627, True, False),
                                                                   (2, True, 12, 4, 29628, False, True), (3, False, 6, 5, 29629, False, True),     # -s / --strand_specific
                                                                   (2, True, 40, 8, 29630, True, True)])
def test_ranks_sharing_one_gpu_equal_single_process(world, paired, n_genes, seed, port, big, ss, tmp_path):
    _ranks_vs_single(world, paired, n_genes, seed, port, big, ss, tmp_path, {})


@pytest.mark.parametrize("world,paired,n_genes,seed,port,big,ss", [(3, True, 12, 4, 29641, False, False), (2, True, 40, 8, 29642, True, False),
                                                                   (2, True, 12, 4, 29643, False, True)])
def test_replicated_table_path_equals_single_process(world, paired, n_genes, seed, port, big, ss, tmp_path):
    """SHN_OWNER_LABELS=0: the path of rounds 2-4 -- the table all-gathered to every rank and labelled there -- kept as the
    comparison for the default (components labelled on the owner shards, no rank holds the whole table)."""
    _ranks_vs_single(world, paired, n_genes, seed, port, big, ss, tmp_path, {"SHN_OWNER_LABELS": "0"})


def _ranks_vs_single(world, paired, n_genes, seed, port, big, ss, tmp_path, env_extra):
    """world_size > 1 with the product's per-rank compute (GpuOps): every rank holds a slice of the reads and runs its HIP
    kernels on cuda:0; the collectives go through gloo (RCCL does not take two ranks on one device).  Covers the sharded
    walks + sharded contig stages, the capped read exchange and partition ownership against the single-process result.
    big: the path of large tables (BASELINE configs[3]: >= 20 M k1-mers) forced with SHN_CONTIG_GPU=1 -- walks sharded by
    component, the candidates of all shards gathered and merged, one replicated GPU contig stage.  ss: -s / --strand_specific
    (forward counting, plain read indices, pairs of reads_1 and RC(reads_2) at the owners)."""
    import json, subprocess, sys
    from conftest import ROOT
    from shannon_amd import device, synth, pipeline, kmers_for_component as kfc
    n_pairs = 12000
    out = str(tmp_path / "res.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    if big:
        env["SHN_CONTIG_GPU"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py"),
                        "1" if paired else "0", str(n_genes), str(seed), str(n_pairs), out] + (["ss"] if ss else []),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    got = json.load(open(out))
    if env_extra.get("SHN_OWNER_LABELS") != "0":
        assert "x:component exchange" in got.get("timings", {}), sorted(got.get("timings", {}))       # the default path ran
        # no rank holds the whole table: the owned shards and the walked tables (whole components) each partition the job's k1-mers
        sizes = got["table_sizes"]
        n_job = got["n_k1mers"]
        assert all(s[2] == n_job for s in sizes)
        assert sum(s[0] for s in sizes) == n_job and sum(s[1] for s in sizes) == n_job
        assert max(s[0] for s in sizes) <= 0.75 * n_job + 1000, sizes
        if n_genes >= 12:                                                       # (several components of size: the walked tables are shares too)
            assert max(s[1] for s in sizes) <= 0.9 * n_job, sizes
    (q1, q2), _ = synth.make_dataset(n_pairs, n_genes, seed=seed)
    if not paired:
        q1, q2 = np.concatenate([q1, q2]), None
    ctx = device.Context(0)
    d1 = device.Reads.from_codes(ctx, q1)
    d2 = device.Reads.from_codes(ctx, q2) if paired else None
    try:
        ref = pipeline.assemble_resident(ctx, d1, d2, kfc.ReadStore(q1, q2), K=25, sample="t", seed=1, double_stranded=not ss)
        assert got["contigs"] == ref.extension.contigs
        assert list(got["partitions"]) == list(ref.partitions)
        for name in ref.partitions:
            assert got["partitions"][name] == ref.partitions[name]["reconstructed_fasta"]
        assert got["final"] == ref.final
    finally:
        d1.close()
        if d2 is not None:
            d2.close()
        ctx.close()


@pytest.mark.parametrize("world,paired,ss", [(2, True, False), (3, False, False), (2, True, True)])
def test_cli_ranks_equal_the_one_process_cli(world, paired, ss, tmp_path):
    """shannon.py -p N: the reference's nJobs (shannon.py:527-566) as N ranks started by the CLI itself before it touches the GPU
    (here sharing cuda:0, collectives over gloo: SHN_CLI_BACKEND=gloo) -- OUT/shannon.fasta and the contig list equal to the
    one-process run of the same command."""
    import subprocess, sys
    from conftest import ROOT
    from shannon_amd import synth
    (q1, q2), _ = synth.make_dataset(12000, 12, seed=4)
    if not paired:
        q1 = np.concatenate([q1, q2])
    f1, f2 = str(tmp_path / "r1.fasta"), str(tmp_path / "r2.fasta")
    synth.write_fasta(f1, q1, "/1")
    if paired:
        synth.write_fasta(f2, q2, "/2")
    files = ["--left", f1, "--right", f2] if paired else ["--single", f1]
    outs = {}
    for tag, extra, env_extra in (("one", [], {}), ("ranks", ["-p", str(world)], {"SHN_CLI_BACKEND": "gloo"})):
        out = str(tmp_path / tag)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "shannon.py"), "-o", out, "-K", "25"] + files + (["-s"] if ss else []) + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **env_extra), timeout=900)
        assert p.returncode == 0, p.stdout[-3000:]
        recs = open(os.path.join(out, "shannon.fasta")).read().split(">")[1:]
        contigs = open(os.path.join(out, "TEMP", tag + "_algo_input", "k1mer.dict_contig")).read()
        outs[tag] = (sorted(r.split("\n", 1)[1] for r in recs), contigs)
        if tag == "ranks":
            assert "%d ranks" % world in p.stdout
    assert outs["one"][1] == outs["ranks"][1]
    assert outs["one"][0] == outs["ranks"][0] and len(outs["one"][0]) > 0
