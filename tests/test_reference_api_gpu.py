"""GPU: the reference's own entry points (shannon_amd/reference_api.py: extension_correction, kmers_for_component,
multibridging.main, algorithm_SF, path_decompose with their signatures and file formats) chained through files exactly as
shannon.py / run_MB_SF_fn.py chain them, against the artefacts of the reference run (tests/golden)."""
import os
import numpy as np
import pytest
from golden_util import *

pytestmark = pytest.mark.gpu


def _canonical_from_files(inter):
    singles, nodes, edges, paths = [], [], [], []
    for l in open(os.path.join(inter, "single_nodes.txt")).read().splitlines()[1:]:
        t = l.split("\t")
        singles.append([t[1], float(t[2]), float(t[3])])
    c = 0
    while os.path.exists(os.path.join(inter, "nodes%d.txt" % c)):
        id2b = {}
        for l in open(os.path.join(inter, "nodes%d.txt" % c)).read().splitlines()[1:]:
            t = l.split("\t")
            id2b[t[0]] = t[1]
            nodes.append([t[1], float(t[2]), float(t[3])])
        for l in open(os.path.join(inter, "edges%d.txt" % c)).read().splitlines()[1:]:
            t = l.split("\t")
            edges.append([id2b[t[0]], id2b[t[1]], int(t[2]), float(t[3]), float(t[4])])
        for l in open(os.path.join(inter, "paths%d.txt" % c)).read().splitlines()[1:]:
            if l.strip():
                paths.append([id2b[x] for x in l.split()])
        c += 1
    return {"single_nodes": sorted(singles), "nodes": sorted(nodes), "edges": sorted(edges), "paths": sorted(paths)}, c


# ds_arg: the `double_stranded` argument of kmers_for_component.  shannon.py passes False in both modes (it is reset at :427): the
# files are then routed as they are -- doubled (default mode) or reads / (reads_1, RC(reads_2)) (-s cases); True: files known to be
# strand-doubled
@pytest.mark.parametrize("name,ds_arg", [("syn_pe_s0", True), ("syn_pe_s0", False), ("syn_se_s7_K20", True), ("syn_se_s7_K20", False),
                                         ("syn_pe_hairpin", True), ("syn_pe_ss_s69", False), ("syn_se_ss_s53", False)])
def test_reference_entry_points_through_files(name, ds_arg, tmp_path):
    from shannon_amd import reference_api as api
    from oracle import seqs, count
    g = load_case(name)
    m = MANIFEST[name]
    K, paired = g["K"], g["paired"]
    inp = load_inputs(name)
    work = str(tmp_path)
    ai = os.path.join(work, "s_algo_input")
    os.makedirs(ai)
    # shannon.py:394-441: strand-doubled read files + the Jellyfish dump (exact counter, KMER-descending: the pinned order)
    dbl = read_files(name, inp)
    rf = []
    for i, reads in enumerate(dbl):
        p = os.path.join(work, "reads_%d.fasta" % (i + 1))
        open(p, "w").write("".join(">%d\n%s\n" % (e, s) for e, s in enumerate(reads)))
        rf.append(p)
    tab = count.count_k1mers_dict([r for f in dbl for r in f], K + 1)
    open(os.path.join(ai, "k1mer.dict_org"), "w").write("".join("%s\t%d\n" % (k, tab[k]) for k in sorted(tab, reverse=True)))
    # shannon.py:450-467
    args = [os.path.join(ai, "k1mer.dict_org"), os.path.join(ai, "k1mer.dict"), "3", "75", work, "500", "1"] + rf
    allowed, reads = api.extension_correction(args, True)
    assert open(os.path.join(ai, "k1mer.dict_contig")).read().split() == g["contigs"]
    assert digest(sorted([k, v] for k, v in allowed.items())) == g["allowed_digest"]
    assert open(os.path.join(work, "reconstructed_single_contigs.fasta")).read() == g["single_contigs_fasta"]
    from shannon_amd import pipeline
    R = pipeline.assemble(api.default_context(), inp[0], inp[1] if paired else None, K=K, sample="p", seed=m["sf_seed"],
                          double_stranded=not strand_specific(name))
    r = api.kmers_for_component(allowed, ai, reads, rf, work, "contigs.txt", True, ds_arg, paired, True, 500, 2, K, "true", 5, False, False, 1)
    assert list(r[1]) == list(g["partitions"])
    for comp, gp in g["partitions"].items():
        files = [os.path.join(work, "reads%s_%s.fasta" % (comp, x)) for x in ("1", "2")] if paired else [os.path.join(work, "reads%s.fasta" % comp)]
        got_reads = [[l.strip() for l in open(f) if l[0] != ">"] for f in files]
        assert digest(got_reads) == gp["reads_digest"]
        assert [l.strip() for l in open(files[0]) if l[0] == ">"][:3] == gp["read_names"]
        kf = os.path.join(work, "component%sk1mers_allowed.dict" % comp)
        assert digest([l.split() for l in open(kf)]) == gp["k1mers_digest"]
        # run_MB_SF_fn.py:219-221, 239-254
        pdir = os.path.join(work, "p_" + comp) + "/"
        arg = "-f --kmer=%d -e --only_k1 %s %s %s %sintermediate" % (K, kf, kf, " ".join(files), pdir)
        api.multibridging_main(arg)
        can, n_comp = _canonical_from_files(pdir + "intermediate")
        for k in can:
            assert approx_eq(can[k], gp["graph"][k])
        api.algorithm_sf(-1, pdir, seed=m["sf_seed"])
        for c in range(n_comp):
            api.algorithm_sf(c, pdir, seed=m["sf_seed"])
        rec = pdir + "algo_output/reconstructed.fasta"
        txt = (open(rec).read() if os.path.exists(rec) else "") + open(pdir + "algo_output/reconstructed_comp_-1.fasta").read()
        # transcripts: those of the in-memory pipeline on the same input (the reference's own transcripts depend on the address
        # order of its node sets -- DESIGN.md 4 -- so the golden text pins one of several outcomes; the graph above is pinned)
        ref, mine = parse_fasta(R.partitions[comp]["reconstructed_fasta"]), parse_fasta(txt)
        assert sorted((sq, h.split("\t")[1]) for h, sq in ref) == sorted((sq, h.split("\t")[1]) for h, sq in mine)


def test_path_decompose_signature_and_closed_forms():
    """path_decompose(a, b, a_true, b_true, overwrite_norm, P, use_GLPK, sparsity) -> [ndarray(m, n), non_unique]
    (path_decompose_sparse.py:15): closed forms (:41-52), marginals of an LP case, the known-answer cases of the reference's
    own wrapper (tests/golden/lp_kats.json, pinned LP rule)."""
    import json
    from shannon_amd import reference_api as api
    x, nu = api.path_decompose([3.0], [1.0, 2.0], [3.0], [1.0, 2.0], 0, np.zeros((1, 2)), False, 10)
    assert x.shape == (1, 2) and x.tolist() == [[1.0, 2.0]] and nu == 0
    x, nu = api.path_decompose([1.0, 2.0], [3.0], None, None, 0, np.zeros((2, 1)), False, 10)
    assert x.tolist() == [[1.0], [2.0]]
    x, nu = api.path_decompose([0.0, 0.0], [1.0, 2.0], None, None, 0, np.zeros((2, 2)), False, 10)
    assert x.tolist() == [[0.0, 0.0], [0.0, 0.0]]
    kats = json.load(open(os.path.join(GOLD, "lp_kats.json")))["kats"]
    from shannon_amd import sparse_flow, device
    ctx = api.default_context()
    for k in kats[:40]:
        m, n = len(k["a"]), len(k["b"])
        kind, *rest = sparse_flow.prepare(k["a"], k["b"], k["P"], k["pid"], k.get("sparsity", 10))
        if kind == "done":
            ans = rest[0]
        else:
            ans, _ = sparse_flow.finish(rest[0], sparse_flow.solve_batch(ctx, [rest[0]], k["seed"])[0])
        assert np.allclose(np.array(ans, dtype=float).reshape(m, n) if m and n else np.zeros((0, 0)),
                           np.array(k["answer"], dtype=float).reshape(m, n) if m and n else np.zeros((0, 0)), rtol=1e-9, atol=1e-12)
