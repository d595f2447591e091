"""GPU parity: partitions, k1-mer emit and read routing (rows a8-a11) vs the reference goldens."""
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


@pytest.mark.parametrize("name", sorted(MANIFEST))
@pytest.mark.parametrize("probe_gpu", ["0", "1"])
def test_routing_matches_golden(ctx, name, probe_gpu, monkeypatch):
    """probe_gpu: k1mers2component by numpy on the host / by the device sort of csrc/probe_gpu.hip"""
    from shannon_amd import device, extension_correction as ec, kmers_for_component as kfc
    monkeypatch.setenv("SHN_PROBE_GPU", probe_gpu)
    g = load_case(name)
    K, paired = g["K"], g["paired"]
    psize = MANIFEST[name].get("partition_size", 500)
    inp = load_inputs(name)
    sets = [device.Reads.from_strings(ctx, r) for r in inp]
    ss = strand_specific(name)
    t = count_case(ctx, name, sets)
    res = ec.run_correction(ctx, t, 3, 75, psize)
    pv = [part_vectors(len(cl), psize) for cl, _ in res.big_components] or None
    for (cl, metis), gb, (p1, p2) in zip(res.big_components, g["big_components"], pv or []):
        assert kfc.weight_updated_graph(metis, p1, 5) == gb["metis_r2"]
    out = kfc.kmers_for_component(ctx, res, sets[0], sets[1] if paired else None, K, psize, part_vectors=pv, strand_specific=ss)
    assert list(out["new_components"]) == list(g["partitions"])
    store = kfc.ReadStore(inp[0], inp[1] if paired else None)
    files = read_files(name, inp)                          # strand-specific: routes are plain read indices into these
    for comp, gp in g["partitions"].items():
        idx = out["routes"][comp]
        assert len(idx) == gp["n_reads"]
        reads = [[files[0][int(d)] for d in idx]] if ss else [[store.mate1(int(d)) for d in idx]]
        if paired:
            reads.append([files[1][int(d)] for d in idx] if ss else [store.mate2(int(d)) for d in idx])
        assert digest(reads) == gp["reads_digest"]
        assert digest([[a, str(b)] for a, b in out["k1mers"][comp]]) == gp["k1mers_digest"]




def test_lazy_routes_equal_downloaded_routes(ctx):
    """kmers_for_component(lazy_routes=True) leaves the routes on the device (RouteView: length, forward-half count and
    slices on demand) -- the same routes as the full download."""
    from shannon_amd import device, synth, extension_correction as ec, kmers_for_component as kfc
    (r1, r2), _ = synth.make_dataset(12000, 12, seed=4)
    d1, d2 = device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)
    t = device.count_k1mers(ctx, [d1, d2], 26, True)
    try:
        res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
        full = kfc.kmers_for_component(ctx, res, d1, d2, 25, 500, want_rows=False)
        lazy = kfc.kmers_for_component(ctx, res, d1, d2, 25, 500, want_rows=False, lazy_routes=True)
        assert list(full["routes"]) == list(lazy["routes"]) and len(full["routes"]) > 0
        for name, r in full["routes"].items():
            v = lazy["routes"][name]
            assert len(v) == len(r) and v.count_below_split() == int(np.searchsorted(r, len(d1)))
            assert np.array_equal(np.asarray(v), r)
            a, b = len(r) // 3, 2 * len(r) // 3
            assert np.array_equal(v[a:b], r[a:b]) and np.array_equal(v[:5], r[:5]) and len(v[len(r):]) == 0
    finally:
        t.close(); d1.close(); d2.close()
