"""CPU: the oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Every stage boundary of the hot path, rows a1-a31."""
import json, os
import pytest
from golden_util import *
from oracle import seqs, count, extension, partition, mbgraph, sparse_flow, lp, post

CASES = sorted(MANIFEST)


def run_oracle_front(name):
    g = load_case(name)
    K, paired = g["K"], g["paired"]
    inp = load_inputs(name)
    dbl = read_files(name, inp)
    tab = count.count_k1mers_dict([r for f in dbl for r in f], K + 1)
    hard = cutoffs(name)[0]
    if hard > 1:                      # `jellyfish dump -L hard` (shannon.py:237-241, 441)
        tab = {k: c for k, c in tab.items() if c >= hard}
    return g, K, paired, dbl, tab


@pytest.mark.parametrize("name", CASES + CUT_CASES)
def test_pipeline_against_reference(name):
    g, K, paired, dbl, tab = run_oracle_front(name)
    psize = meta(name).get("partition_size", 500)
    # a2: k1-mer table as a sorted multiset
    assert len(tab) == g["n_k1mers"]
    assert sum(tab.values()) == g["k1mer_total"]
    assert digest(sorted([k, v] for k, v in tab.items())) == g["k1mer_counts_digest"]
    # packed numpy counter agrees with the dict counter
    keys, cnts = count.count_k1mers_packed([r for f in dbl for r in f], K + 1)
    keys, cnts = keys[cnts >= cutoffs(name)[0]], cnts[cnts >= cutoffs(name)[0]]
    assert len(keys) == len(tab)
    assert all(tab[count.key_to_str(k, K + 1)] == c for k, c in list(zip(keys, cnts))[::97])
    # a3-a7
    items = [(k, tab[k]) for k in sorted(tab, reverse=True)]
    res = extension.run_correction(items, min_weight=cutoffs(name)[1], comp_size_threshold=psize)      # hyp_min_weight: shannon.py:243-247, 457
    assert res.contigs == g["contigs"]
    assert len(res.allowed) == g["n_allowed"]
    assert digest(sorted([k, v] for k, v in res.allowed.items())) == g["allowed_digest"]
    assert "".join(">Single_%d\n%s\n" % (i, c) for i, c in enumerate(res.single_contigs)) == g["single_contigs_fasta"]
    assert res.remaining == g["remaining"]
    assert [b[0] for b in res.big_components] == [b["contigs"] for b in g["big_components"]]
    assert [b[1] for b in res.big_components] == [b["metis"] for b in g["big_components"]]
    # a8-a11
    parts, parts2 = [], []
    for (cl, metis), gb in zip(res.big_components, g["big_components"]):
        p1, p2 = part_vectors(len(cl), psize)
        parts.append(p1)
        parts2.append(p2)
        assert partition.weight_updated_graph(metis, p1, 5) == gb["metis_r2"]
        assert partition.n_partitions(len(cl), psize) == max(p1) + 1
    nc, k2c = partition.build_partitions([b[0] for b in res.big_components], parts, parts2 if parts else None,
                                         res.remaining, res.allowed, K)
    assert list(nc) == list(g["partitions"])
    if paired:
        o1, o2 = partition.route_reads_paired(dbl[0], dbl[1], nc, k2c, K)
    else:
        o1 = partition.route_reads(dbl[0], nc, k2c, K)
    files, cw = partition.partition_k1mers(nc, k2c, K)
    for comp, gp in g["partitions"].items():
        reads = [o1[comp], o2[comp]] if paired else [o1[comp]]
        assert len(reads[0]) == gp["n_reads"]
        assert digest(reads) == gp["reads_digest"]
        rows = [[a, str(b)] for a, b in files[comp]]
        assert digest(rows) == gp["k1mers_digest"]
        # a12-a24
        gr, singles, comps = mbgraph.run_partition(files[comp], reads, K, paired)
        can = mbgraph.canonical(singles, comps)
        for k in can:
            assert approx_eq(can[k], gp["graph"][k]), (comp, k)
        mine_log = [l for l in gr.log]
        ref_log = [l for l in gp["mb_log"] if not l.startswith("0") and l != "Finding known paths."]
        assert [l for l in mine_log if "Bridged" in l] == [l for l in ref_log if "Bridged" in l]


@pytest.mark.parametrize("name", CASES + CUT_CASES)
def test_sparse_flow_against_reference(name):
    """a25-a30 on the reference's own nodes/edges/paths tables (IDs and line order as written)."""
    g = load_case(name)
    seed = meta(name)["sf_seed"]
    for comp, gp in g["partitions"].items():
        mine = ""
        for c, rc in enumerate(gp["raw_components"]):
            tr = sparse_flow.sparse_flow_component(rc["nodes"], rc["edges"], rc["paths"], seed=seed, comp_id=c)
            mine += sparse_flow.fasta_records("", str(c), tr)
        mine += sparse_flow.single_nodes_fasta("", gp["single_rows"])
        ref, mine = parse_fasta(gp["reconstructed_fasta"]), parse_fasta(mine)
        assert len(ref) == len(mine)
        for (h1, s1), (h2, s2) in zip(ref, mine):
            assert s1 == s2
            t1, t2 = h1.split("\t"), h2.split("\t")
            assert t1[0] == t2[0] and t1[2:] == t2[2:]
            if "Copycount" in t1[1]:
                assert t1[1] == t2[1]
            else:
                assert abs(float(t1[1]) - float(t2[1])) <= 1e-9 * max(1.0, abs(float(t1[1])))


def test_lp_kats():
    """a28: wrapper logic of path_decompose against the reference's own wrapper (stub LP/RNG)."""
    kats = json.load(open(os.path.join(GOLD, "lp_kats.json")))["kats"]
    for k in kats:
        ans, nu = lp.path_decompose(k["a"], k["b"], k["P"], seed=k["seed"], pid=k["pid"], sparsity=k.get("sparsity", 10))
        assert approx_eq([list(r) for r in ans], k["answer"], 1e-12), k
        assert nu == k["non_unique"]


def test_lp_matches_highs_objective():
    """The pinned LP rule returns an optimal flow: objective equals scipy-HiGHS (stand-in, not cvxopt)."""
    import numpy as np
    from scipy.optimize import linprog
    rng = np.random.default_rng(3)
    for t in range(60):
        m, n = int(rng.integers(2, 7)), int(rng.integers(2, 7))
        a = rng.integers(1, 20, m).astype(float)
        tot = int(a.sum())
        cuts = np.sort(rng.integers(0, tot + 1, n - 1))
        b = np.diff(np.concatenate([[0], cuts, [tot]])).astype(float)
        ci = [[lp.cell_cost(5, t, 0, j * m + i) * int(rng.random() < 0.7) for j in range(n)] for i in range(m)]
        x = np.array(lp.transport_vertex(list(a), list(b), ci))
        assert np.allclose(x.sum(1), a) and np.allclose(x.sum(0), b) and (x >= 0).all()
        Aeq = []
        for i in range(m):
            r = np.zeros((m, n)); r[i, :] = 1; Aeq.append(r.ravel())
        for j in range(n):
            r = np.zeros((m, n)); r[:, j] = 1; Aeq.append(r.ravel())
        c = np.array(ci, dtype=float) / 2 ** 32
        res = linprog(c.ravel(), A_eq=np.array(Aeq), b_eq=np.concatenate([a, b]), bounds=(0, None), method="highs")
        assert abs(res.fun - (c * x).sum()) < 1e-8


def test_strand_symmetry_and_rc():
    assert seqs.reverse_complement("ATCGGGG") == "CCCCGAT"          # mbgraph.py:69-70 doctest
    g, K, paired, dbl, tab = run_oracle_front("pe_K25")
    for k in list(tab)[::50]:
        assert tab[k] == tab[seqs.reverse_complement(k)]


@pytest.mark.parametrize("name", CASES + CUT_CASES)
def test_final_merge_against_reference(name):
    """a31: the reference's own process_concatenated_fasta -> perl length sort -> faster_reps -d chain (ref_harness.run_final) on
    the concatenation the reference run produced, under both strand settings: names and sequences equal."""
    g = load_case(name)
    lines = g["all_reconstructed"].splitlines(True)
    assert len(lines) >= 2
    for key, ds in (("ds", True), ("ss", False)):
        assert post.finalize(lines, ds) == g["final"][key], key


def test_final_merge_adversarial_against_reference():
    """a31 on the adversarial concatenations of tests/post_cases.py (+-3 containment, reverse-complement duplicates, name-greater
    ties, repeated headers, the 200-base cut), which went through the same reference chain."""
    import hashlib
    from post_cases import adversarial, SEEDS
    g = load_case("post_adversarial")
    for seed in SEEDS:
        lines = adversarial(seed)
        assert hashlib.sha256("".join(lines).encode()).hexdigest() == g[str(seed)]["input_sha256"]
        for key, ds in (("ds", True), ("ss", False)):
            want = g[str(seed)][key]
            assert 20 < len(want) < len(lines) // 2
            assert post.finalize(lines, ds) == want, (seed, key)


@pytest.mark.parametrize("name", ["se_K24", "syn_se_s5", "syn_pe_hairpin", "syn_pe_ss_s69", "syn_part_s33"] + CUT_CASES)
def test_oracle_pipeline_final_against_reference(name):
    """the chained oracle (oracle/pipeline.py) from the input reads to the final file against the reference's final file: the
    sequences (names carry the reference's address-ordered component numbers)"""
    from oracle import pipeline as opipe
    m, g, inp = meta(name), load_case(name), load_inputs(name)
    psize = m.get("partition_size", 500)
    pv = [part_vectors(len(b["contigs"]), psize) for b in g["big_components"]] or None
    ds = not strand_specific(name)
    hard, soft = cutoffs(name)
    O = opipe.assemble(inp[0], inp[1] if m["paired"] else None, K=m["K"], partition_size=psize, sample="", seed=m["sf_seed"],
                       part_vectors=pv, double_stranded=ds, min_weight=soft, kmer_hard_cutoff=hard)
    assert O["contigs"] == g["contigs"]
    assert sorted(O["final"].values()) == sorted(g["final"]["ds" if ds else "ss"].values())


def test_the_cutoff_cases_differ_from_the_default_run():
    """the cutoff fixtures are not the default run under another name: either cutoff changes the contig list of its input"""
    for name in CUT_CASES:
        m = CUT_MANIFEST[name]
        if cutoffs(name) == (1, 3):
            continue
        base = load_case(m.get("default_run", m["input_of"]))
        assert load_case(name)["contigs"] != base["contigs"], name
        if "_se_" not in name:            # (the single-end low-coverage input keeps one transcript of 200 bases either way)
            assert sorted(load_case(name)["final"]["ds"].values()) != sorted(base["final"]["ds"].values()), name
