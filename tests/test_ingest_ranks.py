"""CPU, world 4 (gloo): the N-rank CLI's ingest by BYTES of the read files -- every rank scans and parses its own share only
(shn_text_records_in_range / shn_text_skip_records + shn_reads_ingest on that stretch of text; the reference streams its read files
once: kmers_for_component.py:322-403), the slices of the ranks are the files' records in order, a rank looks at no more than its share
(+ a few records) and holds no more than its slice; and the candidates' gather as tensors (exchange.all_gather_arrays)."""
import json, os, subprocess, sys
import numpy as np
import pytest
from conftest import ROOT


def _write(tmp_path, n, L, fastq, seed, ragged_names=True):
    rng = np.random.default_rng(seed)
    paths = []
    for mate in (1, 2):
        codes = rng.integers(0, 4, (n, L), dtype=np.uint8)
        codes[rng.random((n, L)) < 0.0005] = 4                         # a few N
        A = np.frombuffer(b"ACGTN", np.uint8)
        p = str(tmp_path / ("r%d.%s" % (mate, "fastq" if fastq else "fasta")))
        with open(p, "w") as f:
            for i in range(n):
                name = "read_%d%s/%d" % (i, "_x" * int(rng.integers(0, 6)) if ragged_names else "", mate)      # names of different lengths
                s = A[codes[i]].tobytes().decode()
                if fastq:
                    q = "".join("@+I#"[int(v)] for v in rng.integers(0, 4, L))                                # quality lines that start with '@' or '+'
                    f.write("@%s\n%s\n+\n%s\n" % (name, s, q))
                else:
                    f.write(">%s\n%s\n" % (name, s))
        paths.append(p)
    return paths


def _run(world, out, paths, port):
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "ingest_worker.py"), out] + paths,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]


@pytest.mark.parametrize("fastq,paired,n", [(False, True, 30011), (True, False, 9001), (False, False, 3)])
def test_ranks_ingest_their_share_of_the_bytes(tmp_path, fastq, paired, n):
    from shannon_amd import device
    W, L = 4, 100
    paths = _write(tmp_path, n, L, fastq, seed=3 + n)
    if not paired:
        paths = paths[:1]
    out = str(tmp_path / "o")
    _run(W, out, paths, 29641 + (n % 7))
    whole = [device.Reads.ingest(None, p)[1] for p in paths]
    assert all(w.shape == (n, L) for w in whole)
    total = sum(os.path.getsize(p) for p in paths)
    for r in range(W):
        st = json.load(open("%s.rank%d.json" % (out, r)))
        assert not st.get("declined") and st["n"] == n and st["file_bytes"] == total
        z = np.load("%s.rank%d.npz" % (out, r))
        lo, hi = r * n // W, (r + 1) * n // W
        for i, w in enumerate(whole):
            assert np.array_equal(z["m%d" % i], w[lo:hi]), (r, i)            # the records [n r / W, n (r + 1) / W) of the file, in order
        if n > 1000:
            # a rank looks at its share of the bytes once to count and once to parse (+ the records between a share's start and its
            # slice's): never the whole job
            assert st["bytes_scanned"] <= 1.2 * 2 * total / W, (r, st)
            assert st["bytes_held"] <= 1.2 * (total / W + (hi - lo) * L * len(paths)), (r, st)
        g = json.load(open("%s.gather%d.json" % (out, r)))
        assert g["arrays_equal"]
        assert "test arrays" in g["stats"] and g["stats"]["test arrays"]["bytes_sent"] > 0


def test_files_that_cannot_be_shared_out_are_declined_by_every_rank(tmp_path):
    """reads of different lengths (and .gz files): every rank says so together -- the CLI then reads whole files as before"""
    p = str(tmp_path / "ragged.fasta")
    with open(p, "w") as f:
        for i in range(4000):
            f.write(">r%d\n%s\n" % (i, "ACGT" * (20 + (i % 3))))
    out = str(tmp_path / "o")
    _run(4, out, [p], 29651)
    assert all(json.load(open("%s.rank%d.json" % (out, r))).get("declined") for r in range(4))
