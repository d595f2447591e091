"""shn_reads_ingest, host side (no GPU): the record scan and the code matrix against a line-by-line reading of the same text, as
the reference reads its files (every second line of a 2-line FASTA is a read, rc_gnu.py:15-20; FASTQ: every fourth from the
second) -- FASTA / FASTQ, CRLF, blank lines, no final newline, lower case and N, many byte ranges; what is refused."""
import numpy as np
import pytest

from shannon_amd import _lib, device


def make(n, L, fastq=False, crlf=False, seed=0, qual_at=False):
    rng = np.random.default_rng(seed)
    nl = "\r\n" if crlf else "\n"
    reads, lines = [], []
    for i in range(n):
        s = "".join("ACGTNacgt"[j] for j in rng.choice(9, size=L, p=[.23, .23, .23, .23, .02, .015, .015, .015, .015]))
        reads.append(s)
        if fastq:
            q = "".join(chr(33 + int(x)) for x in rng.integers(0, 40, size=L))
            if qual_at and i % 3 == 0:
                q = "@" + q[1:]                                  # a quality line that looks like a header
            lines += ["@r%d/%d some text" % (i, seed), s, "+" if i % 2 else "+r%d" % i, q]
        else:
            lines += [">read_%d len=%d" % (i * 7919, L), s]
    return nl.join(lines), reads


def expect(reads):
    lut = np.full(256, 4, np.uint8)
    for i, c in enumerate("ACGT"):
        lut[ord(c)] = lut[ord(c.lower())] = i
    return lut[np.frombuffer("".join(reads).encode(), np.uint8)].reshape(len(reads), -1)


@pytest.mark.parametrize("fastq", [False, True])
@pytest.mark.parametrize("crlf,tail", [(False, "\n"), (False, ""), (True, "\r\n"), (False, "\n\n\n")])
@pytest.mark.parametrize("range_bytes", [None, 97, 1000])
def test_ingest_matrix_equals_line_reader(monkeypatch, fastq, crlf, tail, range_bytes):
    if range_bytes:
        monkeypatch.setenv("SHN_INGEST_RANGE_BYTES", str(range_bytes))
    text, reads = make(700, 61, fastq=fastq, crlf=crlf, seed=3, qual_at=True)
    _r, codes = device.Reads.ingest(None, (("\n" if not crlf else "") + text + tail).encode())
    assert codes.shape == (700, 61) and np.array_equal(codes, expect(reads))


def test_ingest_from_files(tmp_path):
    import gzip
    text, reads = make(300, 40, seed=9)
    p = tmp_path / "r.fasta"
    p.write_text(text + "\n")
    with gzip.open(str(p) + ".gz", "wt") as f:
        f.write(text + "\n")
    for path in (str(p), str(p) + ".gz"):
        _r, codes = device.Reads.ingest(None, path)
        assert np.array_equal(codes, expect(reads))


def test_ingest_refuses_what_it_does_not_parse():
    text, reads = make(50, 30, seed=1)
    lines = text.split("\n")
    for bad in ("\n".join(lines[:10] + [lines[11][:12], lines[11][12:]] + lines[12:]),          # a sequence over two lines
                "", "\n\n"):
        with pytest.raises(_lib.ShannonError, match="unsupported"):
            device.Reads.ingest(None, bad.encode())
    q, _ = make(20, 30, fastq=True, seed=2)
    with pytest.raises(_lib.ShannonError, match="unsupported"):
        device.Reads.ingest(None, "\n".join(q.split("\n")[:-1]).encode())                                   # last record without its quality line


def make_ragged(n, fastq=False, crlf=False, seed=0):
    rng = np.random.default_rng(seed)
    nl = "\r\n" if crlf else "\n"
    reads, lines = [], []
    for i in range(n):
        L = int(rng.integers(1, 90)) if i % 11 else 0                 # (an empty sequence line now and then)
        s = "".join("ACGTNacgt"[j] for j in rng.choice(9, size=L, p=[.23, .23, .23, .23, .02, .015, .015, .015, .015]))
        reads.append(s)
        lines += ["@q%d" % i, s, "+", "I" * L] if fastq else [">r%d" % i, s]
    return nl.join(lines), reads


@pytest.mark.parametrize("fastq", [False, True])
@pytest.mark.parametrize("crlf,tail", [(False, "\n"), (False, ""), (True, "\r\n")])
@pytest.mark.parametrize("range_bytes", [None, 53, 4096])
def test_ragged_ingest_equals_line_reader(monkeypatch, fastq, crlf, tail, range_bytes):
    """reads of different lengths (shn_reads_ingest_ragged: the reference's own Samples/SE_read.fasta has 48-51 bases per read):
    codes one after the other + offsets, equal to a line-by-line reading; bases outside ACGT kept as code 4"""
    if range_bytes:
        monkeypatch.setenv("SHN_INGEST_RANGE_BYTES", str(range_bytes))
    text, reads = make_ragged(500, fastq=fastq, crlf=crlf, seed=5)
    _r, rc = device.Reads.ingest(None, (text + tail).encode())
    assert isinstance(rc, device.RaggedCodes) and len(rc) == 500
    assert rc.off.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in reads])]).tolist()
    lut = np.full(256, 4, np.uint8)
    for i, c in enumerate("ACGT"):
        lut[ord(c)] = lut[ord(c.lower())] = i
    assert np.array_equal(rc.codes, lut[np.frombuffer("".join(reads).encode(), np.uint8)])
    assert all(np.array_equal(rc[i], lut[np.frombuffer(reads[i].encode(), np.uint8)]) for i in (0, 1, 11, 250, 499))


def test_read_store_over_ragged_codes_equals_the_string_store():
    """kfc.ReadStore over RaggedCodes (what the CLI hands the graph stage for reads of different lengths): mates as text and the
    gathered codes + strand flags decode to what the store of strings gives, paired and single-end, doubled and strand-specific"""
    from shannon_amd import kmers_for_component as kfc
    rng = np.random.default_rng(2)
    mk = lambda n: ["".join("ACGT"[j] for j in rng.integers(0, 4, size=int(rng.integers(20, 60)))) for _ in range(n)]
    s1, s2 = mk(300), mk(300)

    def rag(strs):
        codes = np.concatenate([np.frombuffer(x.encode(), np.uint8) for x in strs])
        lut = np.zeros(256, np.uint8)
        for i, c in enumerate("ACGT"):
            lut[ord(c)] = i
        off = np.concatenate([[0], np.cumsum([len(x) for x in strs])]).astype(np.uint64)
        return device.RaggedCodes(lut[codes], off)
    comp = str.maketrans("ACGT", "TGCA")
    for paired in (True, False):
        A = kfc.ReadStore(s1, s2 if paired else None)
        B = kfc.ReadStore(rag(s1), rag(s2) if paired else None)
        assert B.ragged and not A.ragged
        idx = np.sort(rng.choice(600, size=200, replace=False))
        for d in idx[::7].tolist():
            assert A.mate1(d) == B.mate1(d) and (not paired or A.mate2(d) == B.mate2(d))
        for mate in ((1, 2) if paired else (1,)):
            for order in (idx, idx[::-1].copy()):
                buf, off, rc, enc = B.gather_codes(order, mate)
                assert enc == 1
                want = [A.mate1(int(d)) if mate == 1 else A.mate2(int(d)) for d in order]
                for j in range(len(order)):
                    t = "".join("ACGT"[c] for c in buf[int(off[j]):int(off[j + 1])])
                    if rc[j]:
                        t = t[::-1].translate(comp)
                    assert t == want[j]
            buf, off, rc, enc = B.gather_codes_ss(idx[idx < 300], mate)
            src = s1 if mate == 1 else s2
            for j, d in enumerate(idx[idx < 300].tolist()):
                t = "".join("ACGT"[c] for c in buf[int(off[j]):int(off[j + 1])])
                assert (t[::-1].translate(comp) if rc[j] else t) == (src[d] if mate == 1 else src[d][::-1].translate(comp))
