"""shn_reads_ingest on the device: a FASTA / FASTQ text ingested (several staging groups) gives the read set, the k1-mer table and
the host code matrix that uploading the parsed reads gives; reads with N are flagged the same."""
import numpy as np
import pytest

from test_ingest import make, expect

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fastq", [False, True])
def test_ingested_reads_count_like_uploaded_ones(monkeypatch, fastq):
    from shannon_amd import device
    monkeypatch.setenv("SHN_INGEST_RANGE_BYTES", "20000")
    monkeypatch.setenv("SHN_INGEST_STAGE_BYTES", "150000")                  # ~15 staging groups
    ctx = device.Context(0)
    text, reads = make(20000, 100, fastq=fastq, seed=11)
    d, codes = device.Reads.ingest(ctx, text.encode())
    want = expect(reads)
    assert np.array_equal(codes, want) and len(d) == len(reads) and d.max_len == 100
    ref = device.Reads.from_codes(ctx, want)
    assert d.n_invalid == ref.n_invalid > 0
    for k1 in (26, 32):
        a = device.count_k1mers(ctx, [d], k1, both_strands=True)
        b = device.count_k1mers(ctx, [ref], k1, both_strands=True)
        ka, ca = a.dump(lower=1)
        kb, cb = b.dump(lower=1)
        assert a.total == b.total and np.array_equal(ka, kb) and np.array_equal(ca, cb)
        a.close(); b.close()
    d.close(); ref.close(); ctx.close()


def test_ragged_reads_with_n_through_the_device_ingest_count_like_uploaded_strings():
    """reads of different lengths with bases outside ACGT (shn_reads_ingest_ragged): the packed set counts like the same reads
    uploaded as strings, and the reference's own Samples/SE_read.fasta (48-51 bases per read) goes through it"""
    import gzip, os
    from shannon_amd import device
    from golden_util import GOLD
    from test_ingest import make_ragged
    ctx = device.Context(0)
    try:
        text, reads = make_ragged(4000, seed=8)
        d, rc = device.Reads.ingest(ctx, (text + "\n").encode())
        assert isinstance(rc, device.RaggedCodes) and len(d) == 4000 and d.n_invalid == sum(1 for r in reads if set(r.upper()) - set("ACGT"))
        want = device.count_k1mers(ctx, [device.Reads.from_strings(ctx, reads)], 21, True)
        got = device.count_k1mers(ctx, [d], 21, True)
        assert got.dump(lower=1)[0].tolist() == want.dump(lower=1)[0].tolist() and got.total == want.total and len(got) > 100
        with gzip.open(os.path.join(GOLD, "data", "SE_read.fasta.gz"), "rb") as f:
            se = f.read()
        d2, rc2 = device.Reads.ingest(ctx, se)
        assert isinstance(rc2, device.RaggedCodes) and len(d2) == 6995 and 48 <= d2.max_len <= 51
    finally:
        ctx.close()
