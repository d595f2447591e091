"""Sequence helpers + strand doubling.  Test infrastructure (see oracle/__init__.py)."""

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def reverse_complement(s):
    """rc_s.py:3-4 / rc_gnu.py:3-4 (A<->T, C<->G, N->N, reversed)."""
    return "".join(_COMP[c] for c in reversed(s))


def double_strand_single(reads):
    """SE strand doubling, shannon.py:396-403: reads.fasta = R ++ RC(R)."""
    return list(reads) + [reverse_complement(r) for r in reads]


def double_strand_paired(r1, r2):
    """PE strand doubling, shannon.py:413-424: reads_1 = R1 ++ RC(R2), reads_2 = RC(R1) ++ R2."""
    return (list(r1) + [reverse_complement(r) for r in r2],
            [reverse_complement(r) for r in r1] + list(r2))


def strand_specific(r1, r2=None):
    """shannon.py:394-424 with -s / --ss / --strand_specific: the reads as they are; of a pair, the second mate reverse-
    complemented (:407-411).  No strand doubling.  Returns the list of read files: [reads] or [reads_1, reads_2]."""
    if r2 is None:
        return [list(r1)]
    return [list(r1), [reverse_complement(s) for s in r2]]


def find_L(reads):
    """rc_gnu.py:15-20: (N, average read length as float)."""
    n = len(reads)
    return n, float(sum(len(r) for r in reads)) / n
