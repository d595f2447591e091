"""GPU: the caching allocator orders the reuse of a freed block behind the stream that may still be writing it (core.hip, round 5).
The hazard of rounds 1-4 -- a block freed while a kernel is queued on one stream goes to a caller on another stream at once -- is
reproduced with the old behaviour behind SHN_DEV_LEGACY=1 (in a child process: the switch is read once), so this test FAILS on
the pre-fix allocator and passes on the current one."""
import os, subprocess, sys
import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
from shannon_amd import _lib, device
L = _lib.lib()
a = device.Context(0)                       # the legacy NULL stream
fh = C.c_void_p()
_lib.check(L.shn_ctx_fork(a.h, C.byref(fh)))   # a second context with a (non-blocking) stream of its own
N = 1 << 22
bad = 0
for it in range(6):
    p = C.c_void_p()
    _lib.check(L.shn_debug_alloc(a.h, N * 4, C.byref(p)))
    _lib.check(L.shn_debug_fill(a.h, p, N, 1, 200000))           # slow: tens of milliseconds on stream A
    L.shn_debug_free(a.h, p)                                      # freed while that kernel is still queued
    q = C.c_void_p()
    _lib.check(L.shn_debug_alloc(fh, N * 4, C.byref(q)))           # the same block, for stream B
    assert q.value == p.value, "the allocator did not hand the freed block on (the test needs it to)"
    _lib.check(L.shn_debug_fill(fh, q, N, 2, 0))                  # fast fill on stream B
    out = np.zeros(N, np.uint32)
    _lib.check(L.shn_debug_read(fh, q, N, out.ctypes.data))
    a.sync()
    _lib.check(L.shn_debug_read(fh, q, N, out.ctypes.data))        # after BOTH streams have drained: B's values must have come last
    bad += int((out != 2).sum())
    L.shn_debug_free(fh, q)
print("WRONG_WORDS", bad, "HANDOVERS", int(L.shn_debug_counter(4)), "DOUBLE", int(L.shn_debug_counter(0)))
p = C.c_void_p()
_lib.check(L.shn_debug_alloc(a.h, 4096, C.byref(p)))
L.shn_debug_free(a.h, p)
L.shn_debug_free(a.h, p)                                          # a second free is reported and counted, the block is not handed out twice
print("DOUBLE_AFTER", int(L.shn_debug_counter(0)))
L.shn_ctx_destroy(fh)
a.close()
""" % ROOT


def _run(**env):
    p = subprocess.run([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env), timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    words = p.stdout.split()
    return {words[i]: int(words[i + 1]) for i in range(0, len(words), 2)}, p.stderr


def test_a_freed_block_is_ordered_behind_the_stream_that_used_it():
    r, err = _run()
    assert r["WRONG_WORDS"] == 0 and r["HANDOVERS"] >= 6, r
    assert r["DOUBLE"] == 0 and r["DOUBLE_AFTER"] == 1 and "freed twice" in err


def test_the_allocator_of_rounds_1_to_4_fails_this():
    """the same sequence on the old allocator (SHN_DEV_LEGACY=1: a freed block goes to the next caller at once): the slow kernel of
    stream A lands on top of stream B's values"""
    r, _err = _run(SHN_DEV_LEGACY="1")
    assert r["WRONG_WORDS"] > 0, r


POISON_CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from golden_util import load_case, load_inputs, MANIFEST
from shannon_amd import device, pipeline
name = sys.argv[1]
m, g, inp = MANIFEST[name], load_case(name), load_inputs(name)
ctx = device.Context(0)
for rep in range(2):
    R = pipeline.assemble(ctx, inp[0], inp[1] if m["paired"] else None, K=m["K"], sample="s", seed=m["sf_seed"])
    assert R.extension.contigs == g["contigs"], "contigs differ under poison (repeat %%d)" %% rep
    assert sorted(R.final.values()) == sorted(g["final"]["ds"].values()), "final differs under poison (repeat %%d)" %% rep
print("POISON_OK", len(R.final))
ctx.close()
""" % (ROOT, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("name", ["syn_pe_s0", "syn_pe_hairpin"])
def test_the_whole_pipeline_with_every_workspace_request_poisoned(name):
    """SHN_DEV_POISON_WS=1 fills a workspace slot with the poison byte on EVERY request: a call that relies on what an earlier call
    left in a slot reads poison.  Until round 5 the seed scan's count / fetch pair did (its offsets lay in the context's slots between
    the two ABI calls: the host crashed, rc 139); the pair owns its offsets now (seeds.hip) and the whole path -- count, walks,
    contig stage, routing, graph threads, LP batches, merge -- gives the reference's contigs and transcripts with poisoned slots
    AND poisoned allocations."""
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", POISON_CHILD, name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, SHN_DEV_POISON="165", SHN_DEV_POISON_WS="1"), timeout=900)
    assert p.returncode == 0 and "POISON_OK" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])


def test_two_pipelines_side_by_side_on_workspace_sets_of_their_own():
    """shn_ctx_own_workspaces (round 6): the stage workspaces -- counting, extension, contig stage -- of a second context in a set
    of its own, so that its stages run BESIDE those of the first context (until then one process-wide set served one pipeline at a
    time: two threads in it overwrite each other's buffers).  Two host threads, a context and a stream each, count + extend
    different inputs at the same time, four times over: every result equals the one made alone."""
    import threading
    import torch
    from shannon_amd import device, synth, extension_correction as ec
    L = __import__("shannon_amd._lib", fromlist=["lib"]).lib()
    own_a, own = torch.cuda.Stream(device=torch.device("cuda", 0)), torch.cuda.Stream(device=torch.device("cuda", 0))
    a = device.Context(0, stream=own_a.cuda_stream)
    b = device.Context(0, stream=own.cuda_stream, own_workspaces=True)
    try:
        sets = []
        for seed, genes in ((31, 40), (32, 25)):
            (r1, r2), _ = synth.make_dataset(60000, genes, seed=seed)
            sets.append((r1, r2))

        def run(ctx, rs):
            d = [device.Reads.from_codes(ctx, rs[0]), device.Reads.from_codes(ctx, rs[1])]
            t = device.count_k1mers(ctx, d, 26)
            res = ec.run_correction(ctx, t, 3, 75, 50)
            out = (len(t), list(res.contigs), {k: list(v) for k, v in res.connections.items()})
            t.close()
            for x in d:
                x.close()
            return out
        alone = [run(a, sets[0]), run(b, sets[1])]
        assert len(alone[0][1]) > 20 and len(alone[1][1]) > 10 and alone[0][1] != alone[1][1]
        before = int(L.shn_debug_counter(2))
        got, errs = [[], []], []

        def worker(i, ctx):
            try:
                for _ in range(4):
                    got[i].append(run(ctx, sets[i]))
            except Exception as ex:                      # noqa: BLE001 -- reported below, from the main thread
                errs.append(repr(ex))
        th = [threading.Thread(target=worker, args=(0, a)), threading.Thread(target=worker, args=(1, b))]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        assert not errs, errs
        assert all(g == alone[0] for g in got[0]) and all(g == alone[1] for g in got[1])
        assert int(L.shn_debug_counter(2)) == before          # nobody asked for a slot of a set whose stage another thread began
    finally:
        b.close()
        a.close()
