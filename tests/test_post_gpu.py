"""Final merge with its two passes over the bases on the device (csrc/post_gpu.hip through shn_post_finalize_dev)
against the oracle's restatement of process_concatenated_fasta.py:6-32, the length sort of shannon.py:603 and
faster_reps.py:60-131 (oracle/post.py) on adversarial inputs: repeated names, exact and reverse-complement duplicates, records
contained in longer ones at every allowed offset (+-3), near-containments just outside it, name-greater ties between equal
lengths, repeated header lines; single- and double-stranded; several text pieces; a base outside ACGT (refused like the host
form)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RC = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s[::-1].translate(RC)


@pytest.fixture(scope="module")
def ctx():
    from shannon_amd import device
    c = device.Context(0)
    yield c
    c.close()


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


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("ds", [True, False])
def test_device_merge_equals_the_oracle(ctx, seed, ds):
    from shannon_amd import post
    from oracle import post as opost
    lines = adversarial(seed)
    want = opost.finalize(lines, ds)
    assert 20 < len(want) < len(lines) // 2
    text = "".join(lines)
    got = post.finalize_texts([text], ds, ctx=ctx)
    assert got == want
    # several pieces (cut at line ends, as the partitions' texts arrive) and an empty one; the host form gives the same
    cuts = [0, len("".join(lines[:100])), len("".join(lines[:101])), len("".join(lines[:700])), len(text)]
    pieces = [text[a:b] for a, b in zip(cuts[:-1], cuts[1:])] + [""]
    assert post.finalize_texts(pieces, ds, ctx=ctx) == want
    assert post.finalize_texts(pieces, ds) == want


def test_device_merge_refuses_other_characters(ctx):
    from shannon_amd import post, _lib
    lines = adversarial(9, 120)
    lines[41] = lines[41][:50] + "N" + lines[41][51:]
    if len(lines[41]) <= 201:
        lines[41] = lines[41].strip() + "ACGT" * 60 + "\n"
    with pytest.raises(_lib.ShannonError, match="non-ACGT"):
        post.finalize_texts(["".join(lines)], True, ctx=ctx)


def test_device_aids_directly(ctx):
    """fingerprints: equal for equal strings, the reverse-complement fingerprint of s is the plain one of rc(s), different lengths
    and one-base changes differ; the scan finds exactly the occurrences a Python scan finds"""
    import ctypes as C
    from shannon_amd import _lib
    rng = np.random.default_rng(3)
    rnd = lambda L: "".join("ACGT"[i] for i in rng.integers(0, 4, size=int(L)))
    a = rnd(700)
    seqs = [a, rc(a), a[:-1], a[:300] + ("A" if a[300] != "A" else "C") + a[301:], rnd(250), a]
    text = "".join(">x%d\n%s\n" % (i, s) for i, s in enumerate(seqs))
    from shannon_amd import post
    out = post.finalize_texts([text], True, ctx=ctx)
    # a, rc(a) and the second a collapse to one record; the one-base variant and the shorter copy are different sequences
    assert sum(1 for v in out.values() if v in (a, rc(a))) == 1
