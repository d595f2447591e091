"""Final merge with its two passes over the bases on the device (csrc/post_gpu.hip through shn_post_finalize_dev)
against the oracle's restatement of process_concatenated_fasta.py:6-32, the length sort of shannon.py:603 and
faster_reps.py:60-131 (oracle/post.py) on adversarial inputs: repeated names, exact and reverse-complement duplicates, records
contained in longer ones at every allowed offset (+-3), near-containments just outside it, name-greater ties between equal
lengths, repeated header lines; single- and double-stranded; several text pieces; a base outside ACGT (refused like the host
form)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from post_cases import adversarial, rc, SEEDS
from golden_util import load_case


@pytest.fixture(scope="module")
def ctx():
    from shannon_amd import device
    c = device.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("ds", [True, False])
def test_device_merge_equals_the_reference_and_the_oracle(ctx, seed, ds):
    from shannon_amd import post
    from oracle import post as opost
    lines = adversarial(seed)
    want = load_case("post_adversarial")[str(seed)]["ds" if ds else "ss"]       # the reference's own chain (ref_harness.run_final)
    assert opost.finalize(lines, ds) == want
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


@pytest.mark.parametrize("seed", SEEDS[:2])
@pytest.mark.parametrize("ds", [True, False])
def test_merge_fed_piece_by_piece_equals_the_reference(ctx, seed, ds):
    """shn_post_stream_*: the pieces handed over in any order, from several threads, each prepared (lines, upload, fingerprints) as it
    arrives; the order-dependent rules over the pieces in index order -- the same survivors as the reference's chain.  The lazy
    mapping over the exported buffers behaves like the dict; fasta() is the final file."""
    from concurrent.futures import ThreadPoolExecutor
    from shannon_amd import post
    lines = adversarial(seed)
    want = load_case("post_adversarial")[str(seed)]["ds" if ds else "ss"]
    cuts = [0, 100, 101, 101, 300, 700, len(lines)]                 # (an empty piece among them)
    pieces = ["".join(lines[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    ps = post.PostStream(ctx, capacity=sum(len(p) for p in pieces) + 64)
    order = [3, 0, 5, 1, 4, 2]
    with ThreadPoolExecutor(max_workers=3) as pool:
        list(pool.map(lambda i: ps.add(i, pieces[i]), order))
    got = ps.finish(ds, lazy=True)
    assert got == want and want == got and len(got) == len(want)
    assert dict(got.items()) == want and sorted(got.values()) == sorted(want.values()) and sorted(got) == sorted(want)
    k = next(iter(want))
    assert got[k] == want[k] and k in got and got.get("no such record") is None
    assert got.fasta().decode() == "".join(">%s\n%s\n" % (n, s) for n, s in got.items())


def test_merge_stream_says_when_it_is_full_or_a_piece_does_not_end_its_line(ctx):
    from shannon_amd import post, _lib
    ps = post.PostStream(ctx, capacity=1 << 20)
    ps.add(0, ">a\n" + "ACGT" * 100 + "\n")
    with pytest.raises(_lib.ShannonError, match="end its last line"):
        ps.add(1, ">b\nACGT")
    with pytest.raises(_lib.ShannonError, match="twice"):
        ps.add(0, ">c\nACGT\n")
    with pytest.raises(_lib.ShannonError, match="full"):
        ps.add(2, ">d\n" + "A" * (2 << 20) + "\n")
    ps.close()
