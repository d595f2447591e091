"""GPU stress: count -> extension -> contig stage and the whole pipeline repeated on the same inputs with the sequential re-walk
audit on (SHN_EXT_AUDIT=2: every walk is re-derived from the converged claims by one thread and the call fails on a difference).
Every repeat must agree with the first one AT EVERY STAGE (shannon_amd/diagnostics.py: the table's multiset and layout, the
checksums shn_extend keeps of its arrays -- weights, adjacency records, seed order, converged claims, walk records -- live stats,
accept filter, emitted contigs, contig stage), and so must run_correction (both pipeline modes) and the final transcripts.  A
failure names the first stage that differs and where; the reference's loop (extension_correction.py:334-397) is sequential, so
there is exactly one right answer per stage."""
import hashlib, json, os
import numpy as np
import pytest
from golden_util import *

pytestmark = pytest.mark.gpu
REPEATS = int(os.environ.get("SHN_STRESS_REPEATS", "20"))


def _sig(res):
    h = hashlib.sha256()
    h.update("\n".join(res.contigs).encode())
    h.update(np.asarray(res.conn_off, np.int64).tobytes() + np.asarray(res.conn_nb, np.int64).tobytes() + np.asarray(res.conn_w, np.int64).tobytes())
    return h.hexdigest()


def _first_contig_difference(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return "contig %d of %d / %d differs (lengths %d / %d)" % (i, len(a), len(b), len(x), len(y))
    return "%d / %d contigs, the common prefix equal" % (len(a), len(b)) if len(a) != len(b) else "contigs equal, connections differ"


@pytest.mark.parametrize("which", ["30genes", "syn_pe_s0"])
def test_repeated_extension_and_pipeline_are_identical_under_audit(which, monkeypatch):
    from shannon_amd import device, synth, extension_correction as ec, pipeline, diagnostics, _lib
    monkeypatch.setenv("SHN_EXT_AUDIT", "2")
    if which == "30genes":
        (r1, r2), _ = synth.make_dataset(40000, 30, seed=17)
        mk = lambda ctx: [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
        A = np.frombuffer(b"ACGT", np.uint8)
        inp = [[A[r].tobytes().decode() for r in r1], [A[r].tobytes().decode() for r in r2]]
        K, seed, pv = 25, 3, None
    else:
        g = load_case(which)
        inp = load_inputs(which)
        mk = lambda ctx: [device.Reads.from_strings(ctx, r) for r in inp]
        K, seed = g["K"], MANIFEST[which]["sf_seed"]
        pv = None
    frees0 = [int(_lib.lib().shn_debug_counter(i)) for i in range(3)]
    ctx = device.Context(0)
    try:
        sets = mk(ctx)
        first_staged = first_ext = first_final = None
        for i in range(REPEATS):
            staged, _info = diagnostics.staged_run(ctx, sets, K)
            if first_staged is None:
                first_staged = staged
            d = diagnostics.first_difference(first_staged, staged)
            if d:
                pytest.fail("repeat %d: the staged run differs from the first one -- %s" % (i, d))
            for pipe in ("1", "0"):                       # contig stage beside the walks / after them
                monkeypatch.setenv("SHN_EXT_PIPELINE", pipe)
                t = device.count_k1mers(ctx, sets, K + 1)
                res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
                t.close()
                s = _sig(res)
                if first_ext is None:
                    first_ext = (s, res.contigs)
                if s != first_ext[0]:
                    pytest.fail("repeat %d (pipeline=%s): run_correction differs from the first run (%s) although the staged run of this repeat agreed"
                                % (i, pipe, _first_contig_difference(first_ext[1], res.contigs)))
            if i % 4 == 0:
                R = pipeline.assemble(ctx, inp[0], inp[1] if len(inp) > 1 else None, K=K, sample="s", seed=seed, part_vectors=pv)
                fin = sorted(R.final.items())
                if first_final is None:
                    first_final = fin
                if fin != first_final:
                    bad = [a[0] for a, b in zip(first_final, fin) if a != b][:3]
                    pytest.fail("repeat %d: final transcripts differ from the first run (%d / %d records, first differing names %s)" % (i, len(first_final), len(fin), bad))
        frees = [int(_lib.lib().shn_debug_counter(i)) for i in range(3)]
        assert frees == frees0, "allocator diagnostics moved: double frees / foreign frees / foreign workspace requests %s -> %s" % (frees0, frees)
    finally:
        ctx.close()
