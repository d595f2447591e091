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
                "\n".join(lines[:21] + [lines[21][:-3]] + lines[22:]),                                # one shorter read
                "", "\n\n"):
        with pytest.raises(_lib.ShannonError, match="unsupported"):
            device.Reads.ingest(None, bad.encode())
    q, _ = make(20, 30, fastq=True, seed=2)
    with pytest.raises(_lib.ShannonError, match="unsupported"):
        device.Reads.ingest(None, "\n".join(q.split("\n")[:-1]).encode())                                   # last record without its quality line
