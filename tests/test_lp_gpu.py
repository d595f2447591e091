"""GPU parity: batched LP trials (row a28) bit-exact vs the oracle's pinned rule, and the whole
sparse-flow stage (rows a25-a30) vs the reference goldens."""
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


def test_lp_batch_bit_exact(ctx):
    from shannon_amd import sparse_flow
    from oracle import lp as olp
    rng = np.random.default_rng(5)
    reqs = []
    for t in range(120):
        m, n = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        if t == 0:
            m, n = 17, 23
        a = [float(v) for v in rng.integers(0, 30, m)]
        if sum(a) == 0:
            a[0] = 3.0
        tot = int(sum(a))
        cuts = np.sort(rng.integers(0, tot + 1, n - 1))
        b = [float(v) for v in np.diff(np.concatenate([[0], cuts, [tot]]))]
        if t % 3 == 0:
            b = [v + float(rng.random()) for v in b]
        P = (rng.random((m, n)) < [0.0, 0.3, 0.8][t % 3]).astype(int).tolist()
        kind, *rest = sparse_flow.prepare(a, b, P, 1000 + t, 10)
        if kind == "lp":
            reqs.append(rest[0])
    xs = sparse_flow.solve_batch(ctx, reqs, 77)
    for q, x in zip(reqs, xs):
        for t in range(q.trials):
            cc = olp.trial_costs(77, q.pid, t, q.m * q.n)
            c = [[(cc[j * q.m + i] if q.p[j * q.m + i] > 0 else 0) for j in range(q.n)] for i in range(q.m)]
            ref = olp.transport_vertex(q.a_s, q.b_s, c)
            flat = np.array([ref[k % q.m][k // q.m] for k in range(q.m * q.n)])
            assert np.array_equal(flat, x[:, t]), (q.m, q.n, t)      # bit-exact
        ans, nu = sparse_flow.finish(q, x)
        # full path_decompose vs oracle
        P = [[1 - int(q.p[j * q.m + i]) for j in range(q.n)] for i in range(q.m)]


def test_path_decompose_kats_gpu(ctx):
    import json, os
    from shannon_amd import sparse_flow
    kats = json.load(open(os.path.join(GOLD, "lp_kats.json")))["kats"]
    for k in kats:
        kind, *rest = sparse_flow.prepare(k["a"], k["b"], k["P"], k["pid"], k.get("sparsity", 10))
        if kind == "done":
            ans, nu = rest
        else:
            x = sparse_flow.solve_batch(ctx, [rest[0]], k["seed"])[0]
            ans, nu = sparse_flow.finish(rest[0], x)
        assert approx_eq([list(r) for r in ans], k["answer"], 1e-12), k
        assert nu == k["non_unique"]


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_sparse_flow_matches_golden(ctx, name):
    from shannon_amd import sparse_flow
    g = load_case(name)
    seed = MANIFEST[name]["sf_seed"]
    for comp, gp in g["partitions"].items():
        comps = [(rc["nodes"], rc["edges"], rc["paths"]) for rc in gp["raw_components"]]
        trs = sparse_flow.sparse_flow_components(ctx, comps, seed)
        mine = "".join(sparse_flow.fasta_records("", str(c), tr) for c, tr in enumerate(trs))
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
                assert abs(float(t1[1]) - float(t2[1])) <= 1e-6 * max(1.0, abs(float(t1[1])))   # north_star: 1e-6 rel


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_native_sparse_flow_equals_python_mirror_and_golden(ctx, name):
    """shn_sparse_flow (native host stage over the LP kernel, all components in one call) on the reference's own raw tables:
    byte-identical FASTA text to the Python mirror of algorithm_SF.py (shannon_amd/sparse_flow.py), and the golden transcripts."""
    from shannon_amd import sparse_flow, mbgraph_native
    g = load_case(name)
    seed = MANIFEST[name]["sf_seed"]
    names, graphs, mirror = [], [], []
    for comp, gp in g["partitions"].items():
        comps = [(rc["nodes"], rc["edges"], rc["paths"]) for rc in gp["raw_components"]]
        trs = sparse_flow.sparse_flow_components(ctx, comps, seed)
        txt = "".join(sparse_flow.fasta_records("s_" + comp, str(c), tr) for c, tr in enumerate(trs))
        txt += sparse_flow.single_nodes_fasta("s_" + comp, gp["single_rows"])
        mirror.append(txt)
        names.append("s_" + comp)
        graphs.append(mbgraph_native.graph_from_tables([tuple(r) for r in gp["single_rows"]],
                                                       [{"nodes": rc["nodes"], "edges": rc["edges"], "paths": rc["paths"]} for rc in gp["raw_components"]]))
    texts = mbgraph_native.sparse_flow_native(ctx, graphs, names, seed)
    for gh in graphs:
        gh.close()
    assert texts == mirror
    for (comp, gp), mine in zip(g["partitions"].items(), texts):
        ref, mine = parse_fasta(gp["reconstructed_fasta"]), parse_fasta(mine)
        assert len(ref) == len(mine)
        for (h1, s1), (h2, s2) in zip(ref, mine):
            assert s1 == s2
            t1, t2 = h1.split("\t"), h2.split("\t")
            assert t1[2:] == t2[2:]
            if "Copycount" not in t1[1]:
                assert abs(float(t1[1]) - float(t2[1])) <= 1e-6 * max(1.0, abs(float(t1[1])))


def test_python_float_repr_of_the_native_stage(ctx):
    """weights are printed with Python's repr(float) rules by the native stage (py_repr in csrc/sflow_host.hip): checked through
    single-node copy counts of many magnitudes"""
    from shannon_amd import mbgraph_native
    vals = [0.5, 1.0, 12.5, 1e-5, 1.5e-5, 0.0001, 123456789.125, 1e15, 1e16, 1.2345678901234567e+22, 3.0000000000000004, 1 / 3.0, 2.5e-7, 65.0, 1e22, 5e-324]
    gh = mbgraph_native.graph_from_tables([(-1, "ACGT", v, 1) for v in vals], [])
    txt = mbgraph_native.sparse_flow_native(ctx, [gh], ["x"], 0)[0]
    gh.close()
    got = [l.split("Copycount:")[1] for l in txt.splitlines() if "Copycount:" in l][1:]
    assert got == [repr(float(v)) for v in vals]
