"""Helpers shared by the parity tests: load golden fixtures and their inputs (data only)."""
import os, json, gzip, hashlib
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
# cases run with non-default --kmer_hard_cutoff / --kmer_soft_cutoff (shannon.py:237-247) on inputs of cases above: a manifest of
# their own, so that `sorted(MANIFEST)` stays "the cases with the default cutoffs"
CUT_MANIFEST = json.load(open(os.path.join(GOLD, "manifest_cutoffs.json")))
CUT_CASES = sorted(CUT_MANIFEST)


def meta(name):
    return MANIFEST[name] if name in MANIFEST else CUT_MANIFEST[name]


def cutoffs(name):
    """(kmer_hard_cutoff, min_weight) of a case: `jellyfish dump -L` and hyp_min_weight (defaults 1, 3: shannon.py:55-56)"""
    m = meta(name)
    return int(m.get("kmer_hard_cutoff", 1)), int(m.get("kmer_soft_cutoff", 3))
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()


def load_case(name):
    return json.load(gzip.open(os.path.join(GOLD, name + ".json.gz"), "rt"))


def load_inputs(name):
    """Returns list of read-string lists: [reads] (SE) or [reads1, reads2] (PE), as in the input files."""
    m = meta(name)
    out = []
    if m["inputs"][0].endswith(".npz"):
        z = np.load(os.path.join(GOLD, "data", m["inputs"][0]))
        mats = [z["r1"], z["r2"]] if m["paired"] else [z["r1"]]
        for mat in mats:
            out.append([ALPHA[row].tobytes().decode() for row in mat])
    else:
        for fn in m["inputs"]:
            with gzip.open(os.path.join(GOLD, "data", fn), "rt") as f:
                out.append([l.strip() for l in f if l.strip() and l[0] != ">"])
    return out


def strand_specific(name):
    """the case was generated with -s / --ss (shannon.py:407-411: no strand doubling, RC(R2) stands for R2)"""
    return bool(meta(name).get("strand_specific"))


DS_CASES = sorted(n for n in MANIFEST if not MANIFEST[n].get("strand_specific"))
SS_CASES = sorted(n for n in MANIFEST if MANIFEST[n].get("strand_specific"))


def read_files(name, inp):
    """the read files the stages after shannon.py:424 see: strand-doubled (default) or as -s leaves them"""
    from oracle import seqs
    paired = meta(name)["paired"]
    if strand_specific(name):
        return seqs.strand_specific(inp[0], inp[1] if paired else None)
    return list(seqs.double_strand_paired(*inp)) if paired else [seqs.double_strand_single(inp[0])]


def count_case(ctx, name, sets):
    """the k1-mer table of a golden case on the device: canonical counting of the doubled input, or -- strand-specific cases --
    forward counting of reads / (reads_1, RC(reads_2))"""
    from shannon_amd import device
    k1 = meta(name)["K"] + 1
    if strand_specific(name):
        t = device.count_k1mers_strand_specific(ctx, sets[0], sets[1] if meta(name)["paired"] else None, k1)
    else:
        t = device.count_k1mers(ctx, sets, k1, both_strands=True)
    hard = cutoffs(name)[0]
    if hard > 1:                      # `jellyfish dump -L hard` (shannon.py:441)
        kept = t.filter_lower(hard)
        t.close()
        t = kept
    return t


def approx_eq(a, b, tol=1e-9):
    if isinstance(a, float) or isinstance(b, float):
        return abs(float(a) - float(b)) <= tol * max(1.0, abs(float(a)), abs(float(b)))
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(approx_eq(x, y, tol) for x, y in zip(a, b))
    return a == b


def parse_fasta(txt):
    ls = txt.splitlines()
    return [(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)]


def part_vectors(n_contigs, psize):
    """The hand-written gpmetis stand-in of tests/golden/make_golden.py (PART_HOOK)."""
    import math
    P = min(int(math.ceil(float(n_contigs) / psize)), 100)
    return [v % P for v in range(n_contigs)], [(v // 2) % P for v in range(n_contigs)]
