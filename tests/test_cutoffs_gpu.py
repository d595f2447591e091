"""GPU parity of the CLI's two k-mer cutoffs with the reference's meaning:

  --kmer_hard_cutoff N = jellyfish_kmer_cutoff, the -L of `jellyfish dump` (shannon.py:237-241, 441): k1-mers counted fewer than N
                         times in the read files the later stages see never enter k1mer.dict_org;
  --kmer_soft_cutoff N = hyp_min_weight -> run_correction's min_weight (shannon.py:243-247, 457): the seed threshold
                         (extension_correction.py:345) and the hyperbola of the accept filter (:361).

The fixtures (tests/golden/cut_*.json.gz, manifest_cutoffs.json) are what the reference itself produced with those settings
(tests/golden/make_golden.py: jellyfish_standin(lower=N), run_correction(min_weight=N)), alone and together, double-stranded and -s.
"""
import os, subprocess, sys
import numpy as np
import pytest
from golden_util import *

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from shannon_amd import device
    c = device.Context(0)
    yield c
    c.close()


def _sets(ctx, inp):
    from shannon_amd import device
    return [device.Reads.from_strings(ctx, r) for r in inp]


@pytest.mark.parametrize("name", ["cut_pe_s0_hard2", "cut_pe_s12_hard3_soft2", "cut_pe_ss_s69_hard2_soft5", "cut_lowcov_s82_hard2_soft2"])
@pytest.mark.parametrize("sk", [False, True])
def test_table_filter_is_jellyfish_dump_L(ctx, name, sk, monkeypatch):
    """shn_table_filter_lower: the filtered table holds exactly the k1-mers of the reference's k1mer.dict_org under `dump -L N` (the
    multiset of the fixture), keeps its bucket structure (every survivor is found by a look-up, every dropped k1-mer is absent) on
    both table layouts (hash buckets / minimizer buckets of the super-k-mer counting path), and leaves the input table as it was."""
    from shannon_amd import device
    from oracle import count
    if sk:
        monkeypatch.setenv("SHN_COUNT_DIRECT", "0")
        monkeypatch.setenv("SHN_COUNT_SK", "2")
    m, g = meta(name), load_case(name)
    hard = cutoffs(name)[0]
    sets = _sets(ctx, load_inputs(name))
    k1 = m["K"] + 1
    ss = strand_specific(name)
    full = (device.count_k1mers_strand_specific(ctx, sets[0], sets[1] if m["paired"] else None, k1) if ss
            else device.count_k1mers(ctx, sets, k1, both_strands=True))
    n_full, total = len(full), full.total
    fk, fc = full.download()
    kept = full.filter_lower(hard)
    assert len(full) == n_full and kept.total == total
    dk, dc = kept.dump(1)                                  # both strands expanded, sorted: the file's multiset
    assert len(dk) == g["n_k1mers"] and int(dc.sum()) == g["k1mer_total"]
    assert digest(sorted([count.key_to_str(int(k), k1), int(c)] for k, c in zip(dk, dc))) == g["k1mer_counts_digest"]
    wk, wc = full.dump(hard)                               # the host rule of shn_table_dump on the unfiltered table: the same pairs
    assert np.array_equal(wk, dk) and np.array_equal(wc, dc)
    # look-ups through the kept table's buckets: every stored key of the old table answers with its count or with 0
    got = kept.lookup(fk)
    pal = np.array([(not ss) and count.rc_key(int(k), k1) == int(k) for k in fk], dtype=bool)
    want = np.where(np.where(pal, 2 * fc.astype(np.int64), fc.astype(np.int64)) >= hard, fc, 0)
    assert np.array_equal(got, want.astype(np.uint32))
    assert len(kept) == int((want > 0).sum()) < n_full
    for s in sets:
        s.close()
    full.close()
    kept.close()


@pytest.mark.parametrize("name", CUT_CASES)
def test_pipeline_with_cutoffs_equals_the_reference(ctx, name):
    """pipeline.assemble(min_weight=soft, kmer_hard_cutoff=hard) against the reference's run with the same two settings: contig
    list, partitions, canonical graphs, transcripts (abundances 1e-6 relative, through the oracle given the same settings) and the
    final file's sequences."""
    from shannon_amd import pipeline, mbgraph
    from oracle import pipeline as opipe
    from test_e2e_gpu import cmp_fasta
    m, g, inp = meta(name), load_case(name), load_inputs(name)
    hard, soft = cutoffs(name)
    ds = not strand_specific(name)
    kw = dict(K=m["K"], partition_size=m.get("partition_size", 500), sample="s", seed=m["sf_seed"], double_stranded=ds, min_weight=soft,
              kmer_hard_cutoff=hard)
    R = pipeline.assemble(ctx, inp[0], inp[1] if m["paired"] else None, **kw)
    O = opipe.assemble(inp[0], inp[1] if m["paired"] else None, **kw)
    assert 0 < R.n_k1mers <= g["n_k1mers"]                                    # (a canonical table holds a strand pair once)
    assert R.extension.contigs == O["contigs"] == g["contigs"]
    assert list(R.partitions) == list(O["partitions"]) == list(g["partitions"])
    for p in R.partitions:
        cmp_fasta(R.partitions[p]["reconstructed_fasta"], O["partitions"][p]["reconstructed_fasta"])
        can = mbgraph.canonical(R.partitions[p]["singles"], R.partitions[p]["components"])
        for k in can:
            assert approx_eq(can[k], g["partitions"][p]["graph"][k])
    assert R.final == O["final"]
    assert sorted(R.final.values()) == sorted(g["final"]["ds" if ds else "ss"].values())


def _write_inputs(tmp_path, name):
    from shannon_amd import synth
    z = np.load(os.path.join(GOLD, "data", meta(name)["inputs"][0]))
    f1, f2 = str(tmp_path / "r1.fasta"), str(tmp_path / "r2.fasta")
    synth.write_fasta(f1, z["r1"], "/1")
    synth.write_fasta(f2, z["r2"], "/2")
    return (["--left", f1, "--right", f2] if meta(name)["paired"] else ["--single", f1])


def _cli(tmp_path, tag, args, env_extra=None):
    from conftest import ROOT
    out = str(tmp_path / tag)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "shannon.py"), "-o", out] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=dict(os.environ, **(env_extra or {})), timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    recs = open(os.path.join(out, "shannon.fasta")).read().split(">")[1:]
    contigs = open(os.path.join(out, "TEMP", tag + "_algo_input", "k1mer.dict_contig")).read().split()
    return sorted(r.split("\n", 1)[1].strip() for r in recs), contigs, p.stdout


@pytest.mark.parametrize("name", ["cut_pe_s0_hard2", "cut_lowcov_s82_soft2", "cut_lowcov_s82_soft5", "cut_lowcov_s82_hard2_soft2",
                                  "cut_pe_ss_s69_hard2_soft5", "cut_lowcov_s82_se_soft5"])
def test_cli_flags_equal_the_reference_run(name, tmp_path):
    """`shannon.py --kmer_hard_cutoff N --kmer_soft_cutoff M` on one rank and on `-p 2` (two ranks sharing cuda:0, collectives over
    gloo): the contig list and OUT/shannon.fasta's sequences equal the reference's run with jellyfish dump -L N and hyp_min_weight M --
    and differ from the reference's default run of the same input."""
    m, g = meta(name), load_case(name)
    hard, soft = cutoffs(name)
    base = load_case(m.get("default_run", m["input_of"]))
    files = _write_inputs(tmp_path, name)
    flags = ["-K", str(m["K"])] + (["-s"] if strand_specific(name) else [])
    if hard != 1:
        flags += ["--kmer_hard_cutoff", str(hard)]
    if soft != 3:
        flags += ["--kmer_soft_cutoff", str(soft)]
    key = "ss" if strand_specific(name) else "ds"
    want = sorted(g["final"][key].values())
    for tag, extra, env in (("one", [], None), ("ranks", ["-p", "2"], {"SHN_CLI_BACKEND": "gloo"})):
        seqs_, contigs, log = _cli(tmp_path, tag, files + flags + extra, env)
        assert contigs == g["contigs"] != base["contigs"], tag
        assert seqs_ == want, tag
        if hard != 1:
            assert "OPTIONS --kmer_hard_cutoff: Kmer hard cutoff set to %d" % hard in log
        if soft != 3:
            assert "OPTIONS --kmer_soft_cutoff: Kmer soft cutoff set to %d" % soft in log
        if tag == "ranks":
            assert "2 ranks" in log
