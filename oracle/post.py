"""Final merge / de-duplication oracle (row a31).  Test infrastructure (see oracle/__init__.py).
Restates process_concatenated_fasta.py:6-32, the perl length sort (shannon.py:603) and
faster_reps.py:60-131.  Record order of the final file is Python-2 dict order in the
reference (faster_reps.py:121) -> compare as a set."""
from .seqs import reverse_complement


def process_concatenated(lines, ds):
    """process_concatenated_fasta.py:6-32.  `lines` include their newlines.  Returns output lines."""
    out = []
    contigs, names_seen = {}, {}
    last_name = ""
    for line in lines:
        tokens = line.split()
        if tokens[0][0] == ">":
            if tokens[0] in names_seen:
                last_name = tokens[0] + "_" + str(names_seen[tokens[0]]) + "\t".join(tokens[1:]) + "\n"
                names_seen[tokens[0]] += 1
            else:
                last_name = line
                names_seen[tokens[0]] = 1
        elif len(line) > 200:                       # includes the newline => seq >= 200 bp
            cur = line.strip()
            if cur in contigs:
                continue
            if ds and reverse_complement(cur) in contigs:
                continue
            contigs[cur] = 1
            out.append(last_name)
            out.append(line)
    return out


def length_sort(lines):
    """shannon.py:603 perl one-liner: records keyed by header line (duplicates collapse, last
    wins), sorted by sequence length ascending (order among equal lengths is perl hash order;
    pinned here: by header)."""
    seqs = {}
    for i in range(0, len(lines) - 1, 2):
        seqs[lines[i]] = lines[i + 1]
    out = []
    for h in sorted(seqs, key=lambda h: (len(seqs[h]), h)):
        out += [h, seqs[h]]
    return out


def find_reps(lines, ds, r=24):
    """faster_reps.py:98-131 + duplicate_check_ends :60-92.  Returns {name: seq} of kept records."""
    contigs, rmer = {}, {}
    name = None
    for line in lines:
        if line[0] == ">":
            name = line.strip().split()[0][1:]
            continue
        seq = line.strip()
        contigs[name] = seq
        for i in range(len(seq) - r + 1):
            rmer.setdefault(seq[i:i + r], []).append([name, i])

    def dup_ends(cname, rc):
        contig = contigs[cname]
        if rc:
            contig = reverse_complement(contig)
        first, last = contig[:r], contig[-r:]
        if first in rmer and last in rmer:
            cd = {}
            for c, p in rmer[first]:
                if c == cname:
                    continue
                if c in cd:
                    cd[c][0] = p
                else:
                    cd[c] = [p, -1]
            for c, p in rmer[last]:
                if c == cname:
                    continue
                if c in cd:
                    cd[c][1] = p
                else:
                    cd[c] = [-1, p]
            for c in cd:
                if cd[c][0] >= 0 and cd[c][1] >= 0:
                    diff = cd[c][1] - cd[c][0]
                    if abs(diff - (len(contig) - r)) < 3:
                        if len(contig) < len(contigs[c]) or (len(contig) == len(contigs[c]) and cname > c):
                            return True
        return False

    kept = {}
    for cname in contigs:
        d = dup_ends(cname, False)
        if ds:
            d = d or dup_ends(cname, True)
        if not d:
            kept[cname] = contigs[cname]
    return kept


def finalize(all_reconstructed_lines, ds=True):
    """shannon.py:596-604: process_concatenated (with the user's strandedness, original_ds) -> length sort -> faster_reps, which
    the reference always calls with -d (:604), whatever the strandedness of the run."""
    return find_reps(length_sort(process_concatenated(all_reconstructed_lines, ds)), True)
