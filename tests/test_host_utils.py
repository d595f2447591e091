"""Host-side helpers of the C ABI that need no GPU: threaded row gather, k-window enumeration."""
import numpy as np
import pytest


def test_gather_rows_matches_numpy_and_checks_its_indices():
    from shannon_amd import _lib, build
    build.build(verbose=False)
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (5000, 100), dtype=np.uint8)
    for n in (0, 1, 7, 40000):
        idx = rng.integers(0, 5000, n)
        assert np.array_equal(_lib.gather_rows(src, idx, threads=8), src[idx])
    with pytest.raises(_lib.ShannonError):
        _lib.gather_rows(src, np.array([5000]))
    with pytest.raises(_lib.ShannonError):
        _lib.gather_rows(src, np.array([-1]))


def test_string_windows_match_the_vectorised_path():
    from shannon_amd import _lib, build
    from shannon_amd.extension_correction import windows_to_keys, windows_to_keys_many
    build.build(verbose=False)
    rng = np.random.default_rng(1)
    strings = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(n))) for n in rng.integers(0, 400, 300)] + ["", "ACG"]
    for k in (1, 15, 26, 32):
        keys, rows, nwin = _lib.string_windows(strings, k, want_keys=True, want_rows=True)
        exp = [windows_to_keys(c, k) for c in strings if len(c) >= k]
        exp = np.concatenate(exp) if exp else np.zeros(0, np.uint64)
        assert np.array_equal(keys, exp) and int(nwin.sum()) == len(exp)
        exp_rows = [np.lib.stride_tricks.sliding_window_view(np.frombuffer(c.encode(), np.uint8), k).reshape(-1) for c in strings if len(c) >= k]
        assert np.array_equal(rows, np.concatenate(exp_rows) if exp_rows else np.zeros(0, np.uint8))
        k2, nw2 = windows_to_keys_many(strings, k)                       # (native above 4096 bases, numpy below)
        assert np.array_equal(k2, exp) and np.array_equal(nw2, nwin)
    with pytest.raises(_lib.ShannonError):
        _lib.string_windows(["ACGTNACGT"], 3)


def _random_fasta(seed, n=400):
    """records with everything process_concatenated / find_reps care about: repeated names, short sequences, repeated and
    reverse-complemented sequences, containments (a prefix / suffix / middle of a longer record), repeated header lines"""
    rng = np.random.default_rng(seed)
    rc = lambda s: s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=int(L))) for L in rng.integers(150, 900, size=n // 2)]
    lines = []
    for i in range(n):
        kind = rng.integers(0, 8)
        base = seqs[int(rng.integers(0, len(seqs)))]
        if kind == 0:
            s = base
        elif kind == 1:
            s = rc(base)
        elif kind == 2 and len(base) > 450:
            a = int(rng.integers(0, 100)); s = base[a:a + 300 + int(rng.integers(0, 50))]
        elif kind == 3 and len(base) > 450:
            s = rc(base[:250 + int(rng.integers(0, 100))])
        elif kind == 4 and len(base) > 450:
            s = base[-(220 + int(rng.integers(0, 100))):]
        else:
            s = "".join("ACGT"[j] for j in rng.integers(0, 4, size=int(rng.integers(150, 700))))
        name = ">s_c%d_%d" % (rng.integers(0, 40), rng.integers(0, 6))
        lines += ["%s\t%.6f\tpath=[%d]\n" % (name, rng.random() * 50, i) if rng.random() < 0.8 else name + "\n", s + "\n"]
    lines += [lines[0], lines[3]]                    # a header line seen twice, with another sequence
    return lines


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("ds", [True, False])
def test_native_merge_equals_the_python_form(seed, ds):
    """shn_post_finalize against process_concatenated + length_sort + find_reps (the readable Python forms of
    process_concatenated_fasta.py:6-32, shannon.py:603, faster_reps.py:60-131)"""
    from shannon_amd import post
    lines = _random_fasta(seed)
    want = post.find_reps(post.length_sort(post.process_concatenated(lines, ds)), True)      # (faster_reps.py always runs with -d, shannon.py:604)
    got = post.finalize_native(lines, ds)
    assert got == want and list(got) == [k for k in got]
    assert 0 < len(got) < len(lines) // 2
    assert post.finalize(lines, ds) == want
    assert post.finalize_native([], ds) == {} and post.finalize_native([">a\n", "ACGT\n"], ds) == {}


def test_native_merge_on_the_host_threads_equals_the_python_form():
    """above 8,192 distinct names find_reps computes its query 24-mers and makes its per-record decisions on the host threads (round
    6: a record's decision reads only the index): 24,000 records with containments, reverse complements and repeated names through
    shn_post_finalize against the sequential Python forms"""
    from shannon_amd import post
    lines = []
    for part in range(12):
        ls = _random_fasta(100 + part, n=2000)
        lines += [(l.replace(">s_c", ">p%d_c" % part, 1) if l[0] == ">" else l) for l in ls]
    want = post.find_reps(post.length_sort(post.process_concatenated(lines, True)), True)
    got = post.finalize_native(lines, True)
    assert len(want) > 8192 and got == want and list(got) == list(want)


def test_allocator_setup_can_be_switched_off():
    """the allocator settings (mallopt) are opt-in: importing the package / loading the library leaves glibc's defaults alone
    (M_MMAP_THRESHOLD: a 4 MB block is mmapped, i.e. malloc_stats' mmap tally grows) unless SHN_MALLOC_TUNE=1 or an explicit
    shannon_amd.malloc_tune() / shn_malloc_tune_now(); either way the import works and the library loads"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, shannon_amd; from shannon_amd import _lib; ok = _lib.host_cpus() >= 1\n"
            "%s\n"
            "libc = ctypes.CDLL(None); libc.mallinfo2.restype = type('MI', (ctypes.Structure,), {'_fields_': [(n, ctypes.c_size_t) for n in "
            "('arena','ordblks','smblks','hblks','hblkhd','usmblks','fsmblks','uordblks','fordblks','keepcost')]})\n"
            "libc.malloc.restype = ctypes.c_void_p; before = libc.mallinfo2().hblks; p = libc.malloc(4 << 20); after = libc.mallinfo2().hblks\n"
            "print(ok, after > before)")
    for env_v, call, mmapped in ((None, "", True), ("0", "", True), ("1", "", False), (None, "shannon_amd.malloc_tune()", False),
                                 (None, "_lib.lib().shn_malloc_tune_now()", False)):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("SHN_MALLOC_TUNE", None)
        for k in ("MALLOC_MMAP_THRESHOLD_", "MALLOC_TRIM_THRESHOLD_", "MALLOC_TOP_PAD_"):
            env.pop(k, None)
        if env_v is not None:
            env["SHN_MALLOC_TUNE"] = env_v
        out = subprocess.run([sys.executable, "-c", code % call], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.split() == ["True", str(mmapped)], (env_v, call, out.stdout, out.stderr)


def test_native_merge_refuses_other_characters_and_the_python_form_takes_over():
    """a transcript with a base outside ACGT: the native merge says so (found by its scan), post.finalize falls back to the Python form"""
    from shannon_amd import post, _lib
    rng = np.random.default_rng(1)
    lines = []
    for i in range(30):
        s = "".join("ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(250, 600))))
        if i == 7:
            s = s[:100] + "N" + s[101:]
        lines += [">t%d\t1.0\n" % i, s + "\n"]
    with pytest.raises(_lib.ShannonError, match="non-ACGT"):
        post.finalize_texts(["".join(lines)], True)
    assert post.finalize(lines, True) == post.find_reps(post.length_sort(post.process_concatenated(lines, True)), True)


def test_native_merge_equals_the_reference_chain():
    """a31, host form of the product (shn_post_finalize_bufs) and its readable Python form against what the reference's own
    process_concatenated_fasta -> perl sort -> faster_reps -d chain produced (tests/golden: ref_harness.run_final): the adversarial
    concatenations and the concatenation of every golden run, under both strand settings"""
    from shannon_amd import post
    from golden_util import load_case, MANIFEST
    from post_cases import adversarial, SEEDS
    g = load_case("post_adversarial")
    for seed in SEEDS:
        lines = adversarial(seed)
        for key, ds in (("ds", True), ("ss", False)):
            assert post.finalize_texts(["".join(lines)], ds) == g[str(seed)][key]
            if seed == 1:
                assert post.find_reps(post.length_sort(post.process_concatenated(lines, ds)), True) == g[str(seed)][key]
    for name in sorted(MANIFEST):
        c = load_case(name)
        for key, ds in (("ds", True), ("ss", False)):
            assert post.finalize_texts([c["all_reconstructed"]], ds) == c["final"][key], (name, key)


def test_final_transcripts_mapping_over_exported_buffers():
    """post.FinalTranscripts: the survivors of the native merge as buffers (names, sequences, offsets) behave like the dict they
    replace -- strings are only made on demand -- and fasta() is the text of the final file"""
    import numpy as np
    from shannon_amd.post import FinalTranscripts
    recs = [("s_c1_0_1", "ACGT" * 60), ("Single_7", "TTGA" * 51 + "C"), ("x", "G" * 201)]
    names = np.frombuffer("".join(n for n, _ in recs).encode(), np.uint8)
    seqs = np.frombuffer("".join(q for _, q in recs).encode(), np.uint8)
    no = np.cumsum([0] + [len(n) for n, _ in recs]).astype(np.uint64)
    so = np.cumsum([0] + [len(q) for _, q in recs]).astype(np.uint64)
    f = FinalTranscripts(names, no, seqs, so)
    want = dict(recs)
    assert len(f) == 3 and list(f) == [n for n, _ in recs] and f.items() == recs and f.values() == [q for _, q in recs]
    assert f == want and want == f and not (f != want) and f != {"x": "G"}
    assert f["Single_7"] == want["Single_7"] and "x" in f and "y" not in f and f.get("y", 5) == 5
    assert f.fasta() == "".join(">%s\n%s\n" % r for r in recs).encode()
    empty = FinalTranscripts(np.zeros(1, np.uint8), np.zeros(1, np.uint64), np.zeros(1, np.uint8), np.zeros(1, np.uint64))
    assert len(empty) == 0 and empty == {} and empty.fasta() == b"" and list(empty) == []


def test_final_transcripts_have_dict_semantics_for_a_repeated_name():
    """faster_reps.py:103-131 keeps its records in a dict keyed by the header's first token: a later record of a name replaces the
    sequence, the name keeps its first place.  post.FinalTranscripts (the lazy view over the native merge's buffers) does the same."""
    import numpy as np
    from shannon_amd import post

    def mk(recs):
        names, seqs = b"".join(n for n, _ in recs), b"".join(s for _, s in recs)
        no = np.concatenate([[0], np.cumsum([len(n) for n, _ in recs])]).astype(np.uint64)
        so = np.concatenate([[0], np.cumsum([len(s) for _, s in recs])]).astype(np.uint64)
        return post.FinalTranscripts(np.frombuffer(names, np.uint8), no, np.frombuffer(seqs, np.uint8), so)
    recs = [(b"a", b"ACGT"), (b"b", b"GG"), (b"a", b"TTTT"), (b"c", b"C"), (b"b", b"GGA")]
    f = mk(recs)
    d = {}
    for k, v in recs:
        d[k.decode()] = v.decode()
    assert len(f) == 3 and f.items() == list(d.items()) and f == d and f["a"] == "TTTT"
    assert f.fasta() == b">a\nTTTT\n>b\nGGA\n>c\nC\n"
    g = mk([(b"x%d" % i, b"ACGT" * (1 + i % 3)) for i in range(500)])          # unique names: untouched
    assert len(g) == 500 and g["x7"] == "ACGT" * 2 and g.fasta().count(b">") == 500


def test_metis_text_of_a_component_equals_the_line_by_line_form():
    """extension_correction.metis_text (one str() pass + a join per line) == the reference's line-by-line formatting
    (extension_correction.py:458-513), rows without connections and members in any order included"""
    import numpy as np
    from shannon_amd import extension_correction as ec
    rng = np.random.default_rng(5)
    for n in (1, 7, 400):
        deg = rng.integers(0, 5, n)
        coff = np.concatenate([[0], np.cumsum(deg)]).astype(np.uint64)
        cnb = rng.integers(1, n + 1, int(deg.sum())).astype(np.int32)
        cw = rng.integers(1, 90, int(deg.sum())).astype(np.int32)
        mm = rng.permutation(np.arange(1, n + 1)).tolist()
        code = {c: i + 1 for i, c in enumerate(mm)}
        o, nb_, w_ = coff.tolist(), cnb.tolist(), cw.tolist()
        want = "%d\t%d\t001\n" % (n, 13) + "".join(
            "".join("%d\t%d\t" % (code[c2], wt) for c2, wt in zip(nb_[o[c - 1]:o[c]], w_[o[c - 1]:o[c]])) + "\n" for c in mm)
        assert ec.metis_text(mm, 13, coff, cnb, cw) == want


def test_partitions_contig_lists_by_one_sort_equal_the_loop():
    """kmers_for_component._contigs_by_part == the reference's loop (kmers_for_component.py:239-262): the same lists under the
    same names IN THE SAME ORDER of first appearance (the partition order decides the order of all_reconstructed.fasta)"""
    import numpy as np
    from shannon_amd import kmers_for_component as kfc
    rng = np.random.default_rng(11)
    for n, P in ((0, 1), (1, 1), (40, 6), (5000, 100)):
        part = rng.integers(0, P, n).tolist()
        contigs = ["c%d" % j for j in range(n)]
        want = {"kept": ["x"]}
        for j, pid in enumerate(part):
            want.setdefault("r2_c%d_%s" % (2, pid), []).append(contigs[j])
        got = {"kept": ["x"]}
        kfc._contigs_by_part(got, "r2_c%d_" % 2, part, contigs)
        assert got == want and list(got) == list(want)
        got2 = {}
        kfc._contigs_by_part(got2, "c1_", np.asarray(part, dtype=np.int32), contigs)      # (a vector as the partitioner returns it)
        assert list(got2.values()) == list(want.values())[1:]


def test_weight_updated_graph_in_one_call_and_with_a_text_that_needs_more_room():
    """shn_metis_reweight through one call with estimated room (round 6) == the Python form; a text whose numbers stand behind
    several blanks and without the trailing tab makes the estimate too small or not -- either way the answer is the same"""
    from shannon_amd import kmers_for_component as kfc
    text = "4\t4\t001\n2\t3\t3\t9\t\n1\t3\t\n1\t9\t4\t1\t\n3\t1\t\n"
    part = [0, 0, 1, 1]
    assert kfc.weight_updated_graph(text, part, 5) == kfc.weight_updated_graph_py(text, part, 5)
    assert kfc.weight_updated_graph(text, part, 123456) == kfc.weight_updated_graph_py(text, part, 123456)
    loose = "4 4 001\n2 3 3 9\n1 3\n1 9 4 1\n3 1\n"
    assert kfc.weight_updated_graph(loose, part, 7) == kfc.weight_updated_graph_py(loose, part, 7)
