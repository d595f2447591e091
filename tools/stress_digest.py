"""Run-to-run determinism of count -> extension -> contig stage with a checksum per stage (GPU).

Every repeat counts the same reads, runs the walks (SHN_EXT_DIGEST=1: shn_extend keeps checksums of its arrays, 8 stages x 64
chunks) and the post-walk path of run_correction step by step (live stats, accept filter, emit, contig stage), hashing what every
step returns.  The first repeat is the reference; a later repeat that differs is reported by the FIRST stage that differs (and,
for the device arrays, the first 1/64 of the array), which is what localises a nondeterminism.  Exit code 1 on a difference.

  python tools/stress_digest.py --repeats 1000 [--case 30genes|syn_pe_s0] [--assemble-every 40] [--pipeline 0|1|both]
"""
import argparse, hashlib, os, sys, time
os.environ["SHN_EXT_DIGEST"] = "1"
os.environ.setdefault("SHN_EXT_AUDIT", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=200)
    ap.add_argument("--case", default="30genes")
    ap.add_argument("--assemble-every", type=int, default=0, help="run the whole pipeline (graph threads on forked streams) every N repeats")
    ap.add_argument("--pipeline", default="both", help="also run ec.run_correction with SHN_EXT_PIPELINE = 0 / 1 / both / none and compare its result")
    ap.add_argument("--seconds", type=float, default=0, help="stop after this many seconds (0: run all repeats)")
    ap.add_argument("--reads", type=int, default=40000)
    ap.add_argument("--genes", type=int, default=30)
    args = ap.parse_args()
    from shannon_amd import device, synth, extension_correction as ec, pipeline, _lib, diagnostics
    if args.case == "30genes":
        (r1, r2), _ = synth.make_dataset(args.reads, args.genes, seed=17)
        A = np.frombuffer(b"ACGT", np.uint8)
        inp = [[A[r].tobytes().decode() for r in r1], [A[r].tobytes().decode() for r in r2]]
        mk = lambda ctx: [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
        K, seed = 25, 3
    else:
        from golden_util import load_case, load_inputs, MANIFEST
        g = load_case(args.case)
        inp = load_inputs(args.case)
        mk = lambda ctx: [device.Reads.from_strings(ctx, r) for r in inp]
        K, seed = g["K"], MANIFEST[args.case]["sf_seed"]
    ctx = device.Context(0)
    sets = mk(ctx)
    first = first_rc = first_final = None
    t0 = time.time()
    bad = 0
    for i in range(args.repeats):
        run, info = diagnostics.staged_run(ctx, sets, K)
        if first is None:
            first = run
            print("repeat 0:", info, flush=True)
        d = diagnostics.first_difference(first, run)
        if d:
            print("REPEAT %d DIFFERS: %s" % (i, d), flush=True)
            bad += 1
        modes = {"both": ("1", "0"), "0": ("0",), "1": ("1",), "none": ()}[args.pipeline]
        for pipe in modes:
            os.environ["SHN_EXT_PIPELINE"] = pipe
            t = device.count_k1mers(ctx, sets, K + 1)
            res = ec.run_correction(ctx, t, 3, 75, 500, want_allowed=False)
            t.close()
            s = diagnostics.h(res.contigs, np.asarray(res.conn_off, np.int64), np.asarray(res.conn_nb, np.int64), np.asarray(res.conn_w, np.int64))
            if first_rc is None:
                first_rc = (s, res.contigs)
            if s != first_rc[0]:
                print("REPEAT %d (pipeline=%s): run_correction differs from the first run: %s" % (i, pipe, diagnostics.describe("contigs", first_rc[1], res.contigs)), flush=True)
                bad += 1
        if args.assemble_every and i % args.assemble_every == 0:
            R = pipeline.assemble(ctx, inp[0], inp[1] if len(inp) > 1 else None, K=K, sample="s", seed=seed)
            fin = sorted(R.final.items())
            if first_final is None:
                first_final = fin
            if fin != first_final:
                print("REPEAT %d: final transcripts differ from the first run" % i, flush=True)
                bad += 1
        if bad >= 5 or (args.seconds and time.time() - t0 > args.seconds):
            break
        if i % 50 == 49:
            print("  %d repeats, %.0f s, %d differences; allocator: double frees %d, foreign frees %d" %
                  (i + 1, time.time() - t0, bad, _lib.lib().shn_debug_counter(0), _lib.lib().shn_debug_counter(1)), flush=True)
    print("DONE repeats=%d differences=%d seconds=%.0f double_frees=%d foreign_frees=%d" %
          (i + 1, bad, time.time() - t0, _lib.lib().shn_debug_counter(0), _lib.lib().shn_debug_counter(1)), flush=True)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
