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
def test_routing_matches_golden(ctx, name):
    from shannon_amd import device, extension_correction as ec, kmers_for_component as kfc
    g = load_case(name)
    K, paired = g["K"], g["paired"]
    psize = MANIFEST[name].get("partition_size", 500)
    inp = load_inputs(name)
    sets = [device.Reads.from_strings(ctx, r) for r in inp]
    t = device.count_k1mers(ctx, sets, K + 1)
    res = ec.run_correction(ctx, t, 3, 75, psize)
    pv = [part_vectors(len(cl), psize) for cl, _ in res.big_components] or None
    for (cl, metis), gb, (p1, p2) in zip(res.big_components, g["big_components"], pv or []):
        assert kfc.weight_updated_graph(metis, p1, 5) == gb["metis_r2"]
    out = kfc.kmers_for_component(ctx, res, sets[0], sets[1] if paired else None, K, psize, part_vectors=pv)
    assert list(out["new_components"]) == list(g["partitions"])
    store = kfc.ReadStore(inp[0], inp[1] if paired else None)
    for comp, gp in g["partitions"].items():
        idx = out["routes"][comp]
        assert len(idx) == gp["n_reads"]
        reads = [[store.mate1(int(d)) for d in idx]]
        if paired:
            reads.append([store.mate2(int(d)) for d in idx])
        assert digest(reads) == gp["reads_digest"]
        assert digest([[a, str(b)] for a, b in out["k1mers"][comp]]) == gp["k1mers_digest"]


