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


def _requests(n_cases=120):
    from shannon_amd import sparse_flow
    rng = np.random.default_rng(5)
    reqs = []
    for t in range(n_cases):
        m, n = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        if t == 0:
            m, n = 17, 23
        if t == 1:
            m, n = 40, 30                      # more than 64 rows + columns: centred as well since round 5 (reachability rows of several words)
        if t == 2:
            m, n = 2, 511                      # more than 512 rows + columns: the vertex is kept (and counted)
        a = [float(v) for v in rng.integers(0, 30, m)]
        if sum(a) == 0:
            a[0] = 3.0
        tot = int(sum(a))
        cuts = np.sort(rng.integers(0, tot + 1, n - 1))
        b = [float(v) for v in np.diff(np.concatenate([[0], cuts, [tot]]))]
        if t % 3 == 0:
            b = [v + float(rng.random()) for v in b]
        P = (rng.random((m, n)) < [0.0, 0.3, 0.8, 1.0][t % 4]).astype(int).tolist()
        kind, *rest = sparse_flow.prepare(a, b, P, 1000 + t, 10)
        if kind == "lp":
            reqs.append(rest[0])
    return reqs


def test_lp_batch_bit_exact(ctx):
    """every trial of 120 random decompositions: the kernels' flows (vertex, then the supported cells at the analytic centre of
    the optimal face) equal oracle/lp.py:transport_center bit for bit; the census counts what the oracle counts"""
    from shannon_amd import sparse_flow
    from oracle import lp as olp
    reqs = _requests()
    ctx.lp_stats(reset=True)
    xs = sparse_flow.solve_batch(ctx, reqs, 77)
    st = ctx.lp_stats()
    assert st["rule"] == "center" and st["lp_calls"] == len(reqs) and st["not_converged"] == 0
    n_deg, n_deg_trials, steps, large = 0, 0, 0, 0
    for q, x in zip(reqs, xs):
        sup = [[not (q.p[j * q.m + i] > 0) for j in range(q.n)] for i in range(q.m)]
        memo, deg = {}, False
        for t in range(q.trials):
            cc = olp.trial_costs(77, q.pid, t, q.m * q.n)
            c = [[(cc[j * q.m + i] if q.p[j * q.m + i] > 0 else 0) for j in range(q.n)] for i in range(q.m)]
            v = olp.transport_vertex(q.a_s, q.b_s, c)
            key = tuple(w for row in v for w in row)
            if key not in memo:
                o = {}
                memo[key] = (olp.face_center(v, sup, o), o)
            ref, o = memo[key]
            n_deg_trials += 1 if o.get("components") else 0
            deg |= bool(o.get("components"))
            steps += o.get("newton_steps", 0)
            large += o.get("too_large", 0)
            flat = np.array([ref[k % q.m][k // q.m] for k in range(q.m * q.n)])
            assert np.array_equal(flat, x[:, t]), (q.m, q.n, t, np.abs(flat - x[:, t]).max())      # bit-exact
        n_deg += deg
    assert (st["lp_degenerate"], st["lp_degenerate_trials"], st["newton_steps"], st["too_large_trials"]) == (n_deg, n_deg_trials, steps, large)
    assert n_deg > 20 and large > 0


def test_wavefront_per_trial_equals_lane_per_trial(ctx, monkeypatch):
    """lp_trials_coop_kernel (a wavefront per trial: the default since round 6) against lp_trials_kernel (lane = trial, state in LDS or
    -- beyond 11 x 11 -- in HBM; SHN_LP_COOP=0): the same bits on every trial, sizes from 2 x 2 to 70 x 3 (rows or columns beyond the
    64 lanes), both rules; the larger ones against the oracle's vertex as well"""
    from shannon_amd import sparse_flow
    from oracle import lp as olp
    rng = np.random.default_rng(11)
    reqs = []
    for t in range(90):
        m, n = int(rng.integers(2, 21)), int(rng.integers(2, 21))
        if t < 4:
            m, n = [(70, 3), (3, 70), (30, 30), (12, 65)][t]
        a = [float(v) for v in rng.integers(0, 30, m)]
        if sum(a) == 0:
            a[0] = 3.0
        tot = int(sum(a))
        cuts = np.sort(rng.integers(0, tot + 1, n - 1))
        b = [float(v) for v in np.diff(np.concatenate([[0], cuts, [tot]]))]
        if t % 3 == 0:
            b = [v + float(rng.random()) for v in b]
        P = (rng.random((m, n)) < [0.0, 0.3, 0.8, 1.0][t % 4]).astype(int).tolist()
        kind, *rest = sparse_flow.prepare(a, b, P, 5000 + t, 10)
        if kind == "lp":
            reqs.append(rest[0])
    assert len(reqs) > 60
    for rule in ("center", "vertex"):
        ctx.set_lp_rule(rule)
        try:
            monkeypatch.setenv("SHN_LP_COOP", "1")
            xa = sparse_flow.solve_batch(ctx, reqs, 91)
            monkeypatch.setenv("SHN_LP_COOP", "0")
            xb = sparse_flow.solve_batch(ctx, reqs, 91)
        finally:
            ctx.set_lp_rule("center")
            monkeypatch.delenv("SHN_LP_COOP", raising=False)
        for q, a_, b_ in zip(reqs, xa, xb):
            assert np.array_equal(a_, b_), (rule, q.m, q.n)
        if rule == "vertex":
            for q, x in list(zip(reqs, xa))[:12]:
                for t in range(0, q.trials, 9):
                    cc = olp.trial_costs(91, q.pid, t, q.m * q.n)
                    c = [[(cc[j * q.m + i] if q.p[j * q.m + i] > 0 else 0) for j in range(q.n)] for i in range(q.m)]
                    ref = olp.transport_vertex(q.a_s, q.b_s, c)
                    assert np.array_equal(np.array([ref[k % q.m][k // q.m] for k in range(q.m * q.n)]), x[:, t])


def test_vertex_rule_behind_the_switch(ctx):
    """shn_lp_set_rule(vertex): the flows are the vertex of rounds 1-2, bit for bit"""
    from shannon_amd import sparse_flow
    from oracle import lp as olp
    reqs = _requests(30)
    ctx.set_lp_rule("vertex")
    try:
        xs = sparse_flow.solve_batch(ctx, reqs, 78)
        assert ctx.lp_stats()["rule"] == "vertex"
    finally:
        ctx.set_lp_rule("center")
    for q, x in zip(reqs, xs):
        for t in range(0, q.trials, 7):
            cc = olp.trial_costs(78, q.pid, t, q.m * q.n)
            c = [[(cc[j * q.m + i] if q.p[j * q.m + i] > 0 else 0) for j in range(q.n)] for i in range(q.m)]
            ref = olp.transport_vertex(q.a_s, q.b_s, c)
            assert np.array_equal(np.array([ref[k % q.m][k // q.m] for k in range(q.m * q.n)]), x[:, t])


def test_whole_path_decompose_equals_the_oracle(ctx):
    """prepare -> kernels -> finish against oracle.lp.path_decompose (thresholds, trial selection, top-10) on the same cases"""
    from shannon_amd import sparse_flow
    from oracle import lp as olp
    rng = np.random.default_rng(11)
    for t in range(60):
        m, n = int(rng.integers(2, 7)), int(rng.integers(2, 7))
        a = [float(v) for v in rng.integers(1, 40, m)]
        b = [float(v) + (float(rng.random()) if t % 2 else 0.0) for v in rng.integers(1, 40, n)]
        P = (rng.random((m, n)) < [0.2, 0.6, 1.0][t % 3]).astype(int).tolist()
        want = olp.path_decompose(a, b, P, seed=5, pid=t, sparsity=10)
        kind, *rest = sparse_flow.prepare(a, b, P, t, 10)
        assert kind == "lp"
        x = sparse_flow.solve_batch(ctx, [rest[0]], 5)[0]
        ans, nu = sparse_flow.finish(rest[0], x)
        assert [list(r) for r in ans] == want[0] and nu == want[1], (t, ans, want)


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
    # ... and on to the final file (a31): the reference's tables through the native sparse flow under the fixture's sample name (""),
    # concatenated behind reconstructed_single_contigs.fasta like shannon.py:584-595 and merged on the device: the reference's own
    # final file (process_concatenated_fasta -> perl sort -> faster_reps -d, ref_harness.run_final), names included
    from shannon_amd import post
    graphs = [mbgraph_native.graph_from_tables([tuple(r) for r in gp["single_rows"]],
                                               [{"nodes": rc["nodes"], "edges": rc["edges"], "paths": rc["paths"]} for rc in gp["raw_components"]])
              for gp in g["partitions"].values()]
    texts = mbgraph_native.sparse_flow_native(ctx, graphs, [""] * len(graphs), seed)
    for gh in graphs:
        gh.close()
    for k2, d2 in (("ds", True), ("ss", False)):
        assert post.finalize_texts([g["single_contigs_fasta"]] + texts, d2, ctx=ctx) == g["final"][k2]


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
