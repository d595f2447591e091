"""Adversarial inputs of the final merge (row a31), shared by the fixture generator (tests/golden/make_golden.py, which pushes them
through the reference's own process_concatenated_fasta -> perl sort -> faster_reps chain) and the parity tests: repeated names,
exact and reverse-complement duplicates, records contained in longer ones at every allowed offset (+-3), near-containments just
outside it, name-greater ties between equal lengths, repeated header lines, lengths around the 200-base cut."""
import numpy as np

RC = str.maketrans("ACGT", "TGCA")
SEEDS = (1, 2, 3, 4)


def rc(s):
    return s[::-1].translate(RC)


def adversarial(seed, n=600):
    rng = np.random.default_rng(seed)
    rnd = lambda L: "".join("ACGT"[i] for i in rng.integers(0, 4, size=int(L)))
    bases = [rnd(L) for L in rng.integers(260, 1500, size=n // 3)]
    lines = []
    for i in range(n):
        kind = int(rng.integers(0, 12))
        b = bases[int(rng.integers(0, len(bases)))]
        if kind == 0:
            s = b                                              # exact duplicate of something else
        elif kind == 1:
            s = rc(b)                                          # reverse-complement duplicate
        elif kind in (2, 3) and len(b) > 600:                  # contained: first and last 24-mer on b at the right distance
            a = int(rng.integers(0, 200)); s = b[a:a + 260 + int(rng.integers(0, 200))]
            if kind == 3:
                s = rc(s)
        elif kind == 4 and len(b) > 600:                       # near-containment: an indel of d bases inside: |diff - (len - 24)| = d
            a = int(rng.integers(0, 100)); t = b[a:a + 400]
            d = int(rng.integers(1, 5))
            s = t[:200] + t[200 + d:] if rng.random() < 0.5 else t[:200] + rnd(d) + t[200:]
        elif kind == 5:                                        # the same length as its container: the name decides
            s = b[:len(b) // 2] + rnd(1) + b[len(b) // 2 + 1:] if rng.random() < 0.3 else b
        elif kind == 6:
            s = rnd(rng.integers(150, 230))                    # around the 200-base cut
        else:
            s = rnd(rng.integers(230, 900))
        name = ">Shannon_s_c%d_%d" % (rng.integers(0, 30), rng.integers(0, 5))
        lines += ["%s\t%.6f\t->S->%d->E\n" % (name, rng.random() * 50, i) if rng.random() < 0.8 else name + "\n", s + "\n"]
    lines += [lines[0], lines[5], lines[2], lines[9]]         # header lines seen twice, with another sequence
    return lines
