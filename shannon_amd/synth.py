"""Deterministic synthetic RNA-Seq generator for BASELINE.json configs 2-5.

Transcriptome: G genes, each 1-6 isoforms assembled from 3-12 exons of 80-600 bp drawn uniform
ACGT, isoform length >= 400 bp; expression ~ LogNormal(0, 1.5); fragments of fixed length 300
with uniform start (isoform chosen with probability ~ expression x #start positions); mates =
first 100 bp of the fragment and reverse complement of its last 100 bp; iid substitution error
0.5 %; no N.  numpy.random.Generator(PCG64(seed)), default seed 20240501.

Base codes everywhere in this package: A=0, C=1, G=2, T=3 (complement = 3 - code; packed
big-endian 2-bit keys order like the strings).
"""
import numpy as np

DEFAULT_SEED = 20240501
ALPHABET = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_transcriptome(n_genes, seed=DEFAULT_SEED, exon_len=(80, 600), n_exons=(3, 12),
                       n_isoforms=(1, 6), min_iso_len=400):
    """Returns (list of uint8 code arrays, one per isoform; list of gene ids)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    isoforms, gene_of = [], []
    for g in range(n_genes):
        ne = int(rng.integers(n_exons[0], n_exons[1] + 1))
        exons = [rng.integers(0, 4, size=int(rng.integers(exon_len[0], exon_len[1] + 1)), dtype=np.uint8)
                 for _ in range(ne)]
        ni = int(rng.integers(n_isoforms[0], n_isoforms[1] + 1))
        seen = set()
        for _ in range(ni):
            for _try in range(20):
                keep = rng.random(ne) < 0.7
                keep[0] = True if ne == 1 else keep[0]
                if not keep.any():
                    continue
                iso = np.concatenate([e for e, k in zip(exons, keep) if k])
                key = keep.tobytes()
                if len(iso) >= min_iso_len and key not in seen:
                    seen.add(key)
                    isoforms.append(iso)
                    gene_of.append(g)
                    break
        if not seen:                      # guarantee at least the full-length isoform
            iso = np.concatenate(exons)
            while len(iso) < min_iso_len:
                iso = np.concatenate([iso, rng.integers(0, 4, size=200, dtype=np.uint8)])
            isoforms.append(iso)
            gene_of.append(g)
    return isoforms, gene_of


def make_repeat_family(n_genes, seed=DEFAULT_SEED, n_repeats=3, repeat_len=120, flank=(200, 400)):
    """Genes that share a few repeated elements (the shape of a transcriptome's repeat-linked super-component): gene i is
    flank + repeat[i % n_repeats] + flank (+ the next repeat in every fourth gene of the class) + tail.  The greedy extension gives the repeat to
    the heaviest gene and cuts the others at its borders, and every contig that overlaps a repeat by a K-mer is connected to its
    contig: ONE component of the contig graph (per strand) with about two contigs per gene -- what sends a component through the
    gpmetis branch (kmers_for_component.py:207-237: components larger than --partition).  BASELINE's configs hold no such component
    (20,000 unrelated genes); `bench.py --config 2p` and the partitioner's pipeline tests are built on this.
    Returns the list of transcript code arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    S = [rng.integers(0, 4, repeat_len, dtype=np.uint8) for _ in range(n_repeats)]
    iso = []
    for i in range(n_genes):
        iso.append(np.concatenate([rng.integers(0, 4, int(rng.integers(flank[0], flank[1])), dtype=np.uint8), S[i % n_repeats],
                                   rng.integers(0, 4, int(rng.integers(flank[0], flank[1])), dtype=np.uint8),
                                   # (every fourth gene of a repeat's class carries the next repeat as well: the classes chain into one component)
                                   S[(i + 1) % n_repeats] if (i // n_repeats) % 4 == 0 else rng.integers(0, 4, 5, dtype=np.uint8),
                                   rng.integers(0, 4, 150, dtype=np.uint8)]))
    return iso


def sample_pairs(isoforms, n_pairs, seed=DEFAULT_SEED, read_len=100, frag_len=300, err=0.005,
                 sigma=1.5, chunk=1 << 20):
    """Returns (r1, r2): uint8 code matrices [n_pairs, read_len]."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    lens = np.array([len(t) for t in isoforms], dtype=np.int64)
    assert (lens >= frag_len).all()
    expr = rng.lognormal(0.0, sigma, size=len(isoforms))
    wts = expr * (lens - frag_len + 1)
    wts /= wts.sum()
    cat = np.concatenate(isoforms)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    r1 = np.empty((n_pairs, read_len), dtype=np.uint8)
    r2 = np.empty((n_pairs, read_len), dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    for s in range(0, n_pairs, chunk):
        n = min(chunk, n_pairs - s)
        iso = rng.choice(len(isoforms), size=n, p=wts)
        start = (rng.random(n) * (lens[iso] - frag_len + 1)).astype(np.int64) + offs[iso]
        a = cat[start[:, None] + ar]
        b = 3 - cat[(start + frag_len - 1)[:, None] - ar]
        for m in (a, b):
            e = rng.random(m.shape) < err
            sub = rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)
            m[e] = (m[e] + sub) & 3
        r1[s:s + n] = a
        r2[s:s + n] = b
    return r1, r2


def codes_to_strings(m):
    return [ALPHABET[row].tobytes().decode() for row in m]


def write_fasta(path, m, suffix=""):
    with open(path, "w") as f:
        for i, row in enumerate(m):
            f.write(">r%d%s\n%s\n" % (i, suffix, ALPHABET[row].tobytes().decode()))


def make_dataset(n_pairs, n_genes, seed=DEFAULT_SEED, **kw):
    iso, _ = make_transcriptome(n_genes, seed)
    return sample_pairs(iso, n_pairs, seed, **kw), iso
