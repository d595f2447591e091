#!/usr/bin/env python3
"""Development aid: random synthetic inputs through the product pipeline (code matrices: reads named by rows, distinct reads and
known_paths on the device, lazy text, native sparse flow and merge) and through the oracle pipeline (oracle/, pure Python, strings);
contigs, partitions and final transcripts must be equal.  python tools/random_parity.py [n_cases] [first_seed] [big]
PARITY_SS=1: every case strand-specific (-s); PARITY_SS=mix: every third."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shannon_amd import device, pipeline, synth, kmers_for_component as kfc
from oracle import pipeline as opipe


A = np.frombuffer(b"ACGT", np.uint8)


def run_case(ctx, seed, big=False):
    """one random case; returns (equal?, description)"""
    if True:
        rng = np.random.default_rng(seed)
        n_pairs = int(rng.integers(1500, 7000)) * (4 if big else 1)
        n_genes = int(rng.integers(3, 16)) * (3 if big else 1)
        K = int(rng.choice([25, 25, 25, 31, 20]))
        paired = bool(rng.integers(0, 2))
        psize = int(rng.choice([4, 8, 500]))
        ss = os.environ.get("PARITY_SS", "") == "1" or (os.environ.get("PARITY_SS", "") == "mix" and seed % 3 == 0)      # -s / --ss (shannon.py:407-411)
        (r1, r2), _ = synth.make_dataset(n_pairs, n_genes, seed=seed)
        s1 = [A[r].tobytes().decode() for r in r1]
        s2 = [A[r].tobytes().decode() for r in r2]
        t = time.time()
        R = pipeline.assemble(ctx, r1, r2 if paired else None, K=K, partition_size=psize, sample="s", seed=seed % 7, double_stranded=not ss)
        tg = time.time() - t
        # components larger than --partition are cut by the library's gpmetis stand-in (kmers_for_component.partition_graph); the
        # oracle takes the cut as an input, like the golden cases do with gpmetis' own output
        pv = []
        for contigs, metis in R.extension.big_components:
            P = kfc.n_partitions(len(contigs), psize)
            p1 = kfc.partition_graph(metis, P, 1000)
            pv.append((p1, kfc.partition_graph(kfc.weight_updated_graph(metis, p1, 5), P, 1000)))
        t = time.time()
        O = opipe.assemble(s1, s2 if paired else None, K=K, partition_size=psize, sample="s", seed=seed % 7, part_vectors=pv or None,
                           double_stranded=not ss)
        to = time.time() - t
        ok = (R.extension.contigs == O["contigs"] and list(R.partitions) == list(O["partitions"]) and R.final == O["final"])
        if ok:
            for p in R.partitions:
                a = [l for l in R.partitions[p]["reconstructed_fasta"].splitlines() if not l.startswith(">")]
                b = [l for l in O["partitions"][p]["reconstructed_fasta"].splitlines() if not l.startswith(">")]
                ok = ok and a == b
        return ok, ("seed %d: pairs %d genes %d K %d %s partition %d: contigs %d partitions %d transcripts %d  %s  (product %.2f s, oracle %.1f s)"
                    % (seed, n_pairs, n_genes, K, ("PE" if paired else "SE") + (" -s" if ss else ""), psize, len(O["contigs"]), len(O["partitions"]), len(O["final"]),
                       "equal" if ok else "DIFFERENT", tg, to))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    ctx = device.Context(0)
    bad = 0
    for c in range(n_cases):
        ok, text = run_case(ctx, seed0 + c, big)
        bad += not ok
        print(text, flush=True)
    print("%d of %d cases differ" % (bad, n_cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
