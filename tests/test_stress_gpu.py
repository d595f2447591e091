"""GPU stress: the extension fixpoint and the whole pipeline repeated on the same inputs with the sequential re-walk audit on
(SHN_EXT_AUDIT=2: every walk is re-derived from the converged claims by one thread and the call fails on a difference).  Every
repeat must give the identical contigs / connections / final transcripts; on a difference the artefacts of both runs are kept
under gpurun_out/stress_fail/ (merged back from the GPU box)."""
import hashlib, json, os
import numpy as np
import pytest
from golden_util import *

pytestmark = pytest.mark.gpu
REPEATS = 20


def _sig(res):
    h = hashlib.sha256()
    h.update("\n".join(res.contigs).encode())
    h.update(np.asarray(res.conn_off, np.int64).tobytes() + np.asarray(res.conn_nb, np.int64).tobytes() + np.asarray(res.conn_w, np.int64).tobytes())
    return h.hexdigest()


def _keep(tag, i, first, now):
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "stress_fail")
    os.makedirs(d, exist_ok=True)
    json.dump({"repeat": i, "first": first, "now": now}, open(os.path.join(d, "%s_%d.json" % (tag, i)), "w"))
    return d


@pytest.mark.parametrize("which", ["30genes", "syn_pe_s0"])
def test_repeated_extension_and_pipeline_are_identical_under_audit(which, monkeypatch):
    from shannon_amd import device, synth, extension_correction as ec, pipeline
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
    ctx = device.Context(0)
    try:
        sets = mk(ctx)
        first_ext = first_final = None
        for i in range(REPEATS):
            for pipe in ("1", "0"):                       # contig stage beside the walks / after them
                monkeypatch.setenv("SHN_EXT_PIPELINE", pipe)
                t = device.count_k1mers(ctx, sets, K + 1)
                res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
                t.close()
                s = _sig(res)
                if first_ext is None:
                    first_ext = (s, res.contigs)
                if s != first_ext[0]:
                    d = _keep(which + "_ext", i, first_ext[1], res.contigs)
                    pytest.fail("repeat %d (pipeline=%s): extension differs from the first run; artefacts in %s" % (i, pipe, d))
            if i % 4 == 0:
                R = pipeline.assemble(ctx, inp[0], inp[1] if len(inp) > 1 else None, K=K, sample="s", seed=seed, part_vectors=pv)
                fin = sorted(R.final.items())
                if first_final is None:
                    first_final = fin
                if fin != first_final:
                    d = _keep(which + "_final", i, first_final, fin)
                    pytest.fail("repeat %d: final transcripts differ from the first run; artefacts in %s" % (i, d))
    finally:
        ctx.close()
