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
