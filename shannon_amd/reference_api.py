"""The reference's own Python entry points for the hot path, with their signatures, argument meaning and file formats
(SURVEY.md 8b), over the MI355X library -- what a maintainer of sreeramkannan/Shannon would import instead of the
modules of the same names:

    extension_correction(arguments, inMem)                      extension_correction.py:528
    kmers_for_component(... 18 arguments ...)                    kmers_for_component.py:144
    multibridging_main(arguments, inMem, contigs, weights, rps)  multibridging.py:327 (`main`)
    algorithm_sf(comp, prefix)                                   algorithm_SF.py:39-63, 858-934 (the script)
    path_decompose(a, b, a_true, b_true, overwrite_norm, P, use_GLPK, sparsity)   path_decompose_sparse.py:15

They read and write the reference's files (k1mer.dict_org / *_contig / component*.txt / remaining_contigs*.txt /
reads{comp}.fasta / component{comp}k1mers_allowed.dict / nodes, edges, paths{c}.txt / reconstructed*.fasta).  The pipeline
(shannon_amd/pipeline.py) hands the same data over in memory; these wrappers exist so that a single stage can be swapped
into a reference run.  No CPU fallback: every wrapper drives the HIP library (a device context on GPU 0 by default).
"""
import os
import sys
import numpy as np
from . import device, extension_correction as ec, kmers_for_component as kfc, mbgraph, mbgraph_native, sparse_flow

_ctx = [None]


def default_context():
    if _ctx[0] is None:
        _ctx[0] = device.Context(0)
    return _ctx[0]


def _read_fasta(path):
    seqs, cur = [], None
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if cur is not None:
                    seqs.append(cur)
                cur = ""
            elif cur is not None:
                cur += line.strip()
    if cur is not None:
        seqs.append(cur)
    return seqs


def _table_from_dump(ctx, path):
    """k1mer.dict_org (`jellyfish dump -c -t`: KMER<TAB>count, every k1-mer of the strand-doubled input) -> device table"""
    keys, counts, k1 = [], [], None
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 2:
                continue
            k1 = len(t[0])
            keys.append(device.str_to_key(t[0]))
            counts.append(int(float(t[1])))
    if k1 is None:
        raise ValueError("%s holds no k1-mers" % path)
    return kfc.make_table(ctx, np.array(keys, np.uint64), np.array(counts, np.uint32), k1, canonical=False), k1


def extension_correction(arguments, inMem=False, ctx=None):
    """extension_correction.py:528-548.  arguments = [infile, outfile, min_weight, min_length, comp_directory_name,
    comp_size_threshold, nJobs, reads_file(s)...] (+ '-d'); writes `outfile` (allowed k1-mers with weights), `outfile`_contig,
    reconstructed_single_contigs.fasta, component{i}.txt / component{i}contigs.txt, remaining_contigs{i}.txt into
    comp_directory_name (:337, 362, 458-513).  Returns (allowed_kmer_dict, reads) like the reference (reads: the read files'
    sequences when inMem and read files are given, else [])."""
    ctx = ctx or default_context()
    arguments = [a for a in arguments if len(a) > 0 and a[0] != "-"]
    infile, outfile = arguments[:2]
    min_weight, min_length = int(arguments[2]), int(arguments[3])
    comp_directory_name, comp_size_threshold = arguments[4], int(arguments[5])
    reads_files = arguments[7:9] if len(arguments) > 7 else []
    table, _k1 = _table_from_dump(ctx, infile)
    try:
        res = ec.run_correction(ctx, table, min_weight, min_length, comp_size_threshold, want_allowed=True)
    finally:
        table.close()
    ec.write_outputs(res, comp_directory_name, outfile)
    if not inMem:
        with open(outfile, "w") as f:
            for kmer, w in res.allowed.items():
                f.write("%s\t%d\n" % (kmer, w))
    reads = [_read_fasta(p) for p in reads_files] if (inMem and reads_files) else []
    return res.allowed, reads


def kmers_for_component(k1mer_dictionary, kmer_directory, reads, reads_files, directory_name, contig_file_extension, get_partition_k1mers,
                        double_stranded=True, paired_end=False, repartition=False, partition_size=500, overload=1.5, K=24,
                        gpmetis_path="gpmetis", penalty=5, only_reads=False, inMem=False, nJobs=1, ctx=None, part_vectors=None):
    """kmers_for_component.py:144-558.  Reads component{i}contigs.txt + component{i}.txt (METIS) and remaining_contigs{i}.txt from
    directory_name, partitions the big components (gpmetis is replaced by the deterministic partitioner of
    shannon_amd/kmers_for_component.py; `part_vectors` replays given gpmetis vectors), routes the reads of reads_files (as
    shannon.py:394-424 wrote them; double_stranded=False -- what shannon.py passes -- routes them as they are) and writes reads{comp}.fasta (paired: reads{comp}_1.fasta / _2.fasta) and
    component{comp}k1mers_allowed.dict.  Returns [components_broken, new_comps, contig_weights, rps] (:558)."""
    ctx = ctx or default_context()
    res = ec.ExtensionResult()
    res.allowed = dict(k1mer_dictionary)
    res.big_components, res.remaining = [], []
    i = 1
    while os.path.exists(os.path.join(directory_name, "component%dcontigs.txt" % i)):
        res.big_components.append((open(os.path.join(directory_name, "component%dcontigs.txt" % i)).read().split(),
                                   open(os.path.join(directory_name, "component%d.txt" % i)).read()))
        i += 1
    i = 1
    while os.path.exists(os.path.join(directory_name, "remaining_contigs%d.txt" % i)):
        res.remaining.append(open(os.path.join(directory_name, "remaining_contigs%d.txt" % i)).read().split())
        i += 1
    files = [_read_fasta(p) for p in reads_files] if reads_files else [list(r) for r in reads]
    if not double_stranded:
        # what shannon.py really passes (double_stranded is False from :427 on, in both modes): the files are routed as they are --
        # strand-doubled by :394-424 in the default mode, reads / (reads_1, RC(reads_2)) with -s -- read i of a pair file with
        # read i of the other, forward k1-mers only (kmers_for_component.py:186-205, 322-403).  shn_route_reads_mode(strand_specific)
        # takes the reverse complement of the second file's reads itself: it is handed their reverse complements.
        r1 = files[0]
        r2 = [kfc.ReadStore._rc(q) for q in files[1]] if paired_end else None
        d1 = device.Reads.from_strings(ctx, r1)
        d2 = device.Reads.from_strings(ctx, r2) if r2 is not None else None
        out = kfc.kmers_for_component(ctx, res, d1, d2, K, partition_size, overload, penalty, repartition, part_vectors, want_rows=True,
                                      strand_specific=True)
        mates = lambda idx: ([files[0][int(d)] for d in idx], [files[1][int(d)] for d in idx] if paired_end else None)
    else:
        # double_stranded=True: the read files hold the strand-doubled reads (the first half are the reads as given, the second half
        # the other strand) and are un-doubled for the kernels, which double on chip
        n_half = len(files[0]) // 2
        if paired_end:
            # reads_1 = R1 ++ RC(R2), reads_2 = RC(R1) ++ R2 (shannon.py:413-424)
            r1, r2 = files[0][:n_half], files[1][n_half:]
        else:
            r1, r2 = files[0][:n_half], None
        d1 = device.Reads.from_strings(ctx, r1)
        d2 = device.Reads.from_strings(ctx, r2) if r2 is not None else None
        out = kfc.kmers_for_component(ctx, res, d1, d2, K, partition_size, overload, penalty, repartition, part_vectors, want_rows=True)
        store = kfc.ReadStore(r1, r2)
        mates = lambda idx: ([store.mate1(int(d)) for d in idx], [store.mate2(int(d)) for d in idx] if paired_end else None)
    d1.close()
    if d2 is not None:
        d2.close()
    rps = {}
    for comp, idx in out["routes"].items():
        m1, m2 = mates(idx)
        if inMem:
            rps[comp] = [m1, m2] if paired_end else [m1]
        else:
            for suffix, lst in ((("_1", m1), ("_2", m2)) if paired_end else (("", m1),)):
                with open(os.path.join(directory_name, "reads%s%s.fasta" % (comp, suffix)), "w") as f:
                    for e, s in enumerate(lst):
                        f.write(">%d%s\n%s\n" % (e, suffix, s))              # '>e_1' / '>e_2' (kmers_for_component.py:396-397), '>e' (:349)
    if get_partition_k1mers and not inMem:
        for comp, rows in out["k1mers"].items():
            with open(os.path.join(directory_name, "component%sk1mers_allowed.dict" % comp), "w") as f:
                for km, w in rows:
                    f.write("%s\t%d\n" % (km, w))
    new_comps = list(out["new_components"]) if not inMem else out["new_components"]
    return [out["components_broken"], new_comps, out["contig_weights"] if inMem else {}, rps]


def multibridging_main(arguments, inMem=False, contigs=(), weights=(), rps=(), ctx=None):
    """multibridging.main (multibridging.py:327-400): `-f --kmer=K -e --only_k1 <kmer.dict> <k1mer.dict> <reads file(s)> <output_dir>`
    -- loads the partition's k1-mers and reads, runs the multibridged graph stage and writes single_nodes.txt and
    nodes / edges / paths{c}.txt into output_dir (:271-325).  inMem: `contigs` = the partition's contig strings, `rps` = [reads]
    or [reads_1, reads_2] (`weights` is not needed: the graph takes no k1-mer weight, multibridging.py:169 quirk)."""
    ctx = ctx or default_context()
    args = arguments.strip().split()
    K, names = 24, []
    for a in args:
        if a.startswith("--kmer="):
            K = int(a[7:])
        elif not a.startswith("-"):
            names.append(a)
    output_dir = names[-1]
    if inMem:
        rows = [(c[i:i + K + 1], 0) for c in contigs for i in range(len(c) - K)]
        reads = [list(r) for r in rps]
    else:
        edge_file, read_files = names[1], names[2:-1]
        rows = [(l.split()[0], 0) for l in open(edge_file) if l.strip()]
        reads = [_read_fasta(p) for p in read_files]
    paired = len(reads) == 2
    cutoff = 10 * _n_kmers(rows, K) + 1                            # multibridging.py:26-30, 385-391
    reads = [r[:cutoff] for r in reads]
    singles, comps, _log = mbgraph_native.run_partition(rows, reads, K, paired, ctx=ctx)
    os.makedirs(output_dir, exist_ok=True)
    mbgraph.write_files(singles, comps, output_dir)
    return None


def _n_kmers(rows, K):
    s = set()
    for km, _ in rows:
        s.add(km[:-1])
        s.add(km[1:])
    return len(s)


def _read_tables(prefix, comp):
    def rd(path):
        return [l.split("\t") for l in open(path).read().splitlines()[1:] if l.strip()]
    inter = os.path.join(prefix + "intermediate")
    if comp == -1 or comp == "-1":
        rows = [(int(t[0]), t[1], (0 if t[2] in ("0", "0.0") else float(t[2])), int(float(t[3]))) for t in rd(os.path.join(inter, "single_nodes.txt"))]
        return rows, None
    nodes = [(int(t[0]), t[1], (0 if t[2] == "0" else float(t[2])), int(float(t[3]))) for t in rd(os.path.join(inter, "nodes%s.txt" % comp))]
    edges = [(int(t[0]), int(t[1]), int(t[2]), float(t[3]), int(float(t[4]))) for t in rd(os.path.join(inter, "edges%s.txt" % comp))]
    paths = [[int(x) for x in t] for t in rd(os.path.join(inter, "paths%s.txt" % comp))]
    return None, {"nodes": nodes, "edges": edges, "paths": paths}


def algorithm_sf(comp, prefix, seed=0, ctx=None):
    """`python algorithm_SF.py <comp> <prefix>` (algorithm_SF.py:39-63, 858-934; run_MB_SF_fn.py:239-250): reads
    <prefix>intermediate/{nodes,edges,paths}<comp>.txt (comp = -1: single_nodes.txt) and appends the component's transcripts to
    <prefix>algo_output/reconstructed.fasta (comp = -1: writes reconstructed_comp_-1.fasta)."""
    ctx = ctx or default_context()
    sname = os.path.basename(os.path.normpath(prefix.rstrip("/")))
    outdir = prefix + "algo_output"
    os.makedirs(outdir, exist_ok=True)
    singles, comp_tables = _read_tables(prefix, comp)
    if comp_tables is None:
        txt = sparse_flow.single_nodes_fasta(sname, singles)
        open(os.path.join(outdir, "reconstructed_comp_-1.fasta"), "w").write(txt)
        return txt
    trs = sparse_flow.sparse_flow_components(ctx, [(comp_tables["nodes"], comp_tables["edges"], comp_tables["paths"])], seed)[0]
    txt = sparse_flow.fasta_records(sname, str(comp), trs)
    with open(os.path.join(outdir, "reconstructed.fasta"), "a") as f:
        f.write(txt)
    return txt


def path_decompose(a, b, a_true, b_true, overwrite_norm, P, use_GLPK=False, sparsity=False, seed=0, ctx=None):
    """path_decompose_sparse.py:15-193: decomposes the flows a (in-edges) and b (out-edges) of a node into an m x n flow matrix of
    few non-zero cells outside the support P.  Returns [ndarray(m, n), non_unique].  a_true / b_true / overwrite_norm / use_GLPK
    are accepted and unused, as in the reference's live code path.  cvxopt is not a dependency: the trial LPs return the limit of
    its interior-point method (exact vertex flows on the unsupported cells, analytic centre of the optimal face on the supported
    ones; csrc/lp.hip == oracle/lp.py bit for bit), costs from a counter-based stream seeded by `seed` and the call number."""
    ctx = ctx or default_context()
    m, n = len(a), len(b)
    Pm = [[int(round(float(P[i, j] if hasattr(P, "shape") else P[i][j]))) for j in range(n)] for i in range(m)] if m and n else []
    # (the call number that seeds the cost stream is kept per context, not in a module global: contexts may work side by side)
    call = getattr(ctx, "_pd_calls", 0)
    ctx._pd_calls = call + 1
    kind, *rest = sparse_flow.prepare([float(v) for v in a], [float(v) for v in b], Pm, call, int(sparsity) if sparsity else 0)
    if kind == "done":
        ans = rest[0]
        return [np.array(ans, dtype=float).reshape(m, n) if m and n else np.zeros((0, 0)), rest[1] if len(rest) > 1 else 0]
    q = rest[0]
    xs = sparse_flow.solve_batch(ctx, [q], seed)[0]
    ans, non_unique = sparse_flow.finish(q, xs)
    return [np.array(ans, dtype=float).reshape(m, n), non_unique]
