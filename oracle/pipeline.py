"""End-to-end oracle: chains the stage oracles in the order of shannon.py:394-647.  Test
infrastructure (see oracle/__init__.py)."""
from . import seqs, count, extension, partition, mbgraph, sparse_flow, post


def assemble(reads1, reads2=None, K=25, partition_size=500, sample="shannon", seed=0, part_vectors=None, double_stranded=True,
             min_weight=3, min_length=75, kmer_hard_cutoff=1):
    """double_stranded=False: the -s / --ss run -- after the read files are made the reference sets double_stranded = False for
    every later stage in both modes (shannon.py:427); only the read files differ (:394-424), and process_concatenated_fasta at
    the end, which gets the user's flag (:596).
    min_weight / min_length: hyp_min_weight (--kmer_soft_cutoff) / hyp_min_length as run_correction gets them (shannon.py:243-247,
    457); kmer_hard_cutoff: `jellyfish dump -L` (shannon.py:237-241, 441) -- k1-mers counted fewer times in the read files the
    later stages see are not in k1mer.dict_org."""
    paired = reads2 is not None
    if double_stranded:
        dbl = list(seqs.double_strand_paired(reads1, reads2)) if paired else [seqs.double_strand_single(reads1)]
    else:
        dbl = seqs.strand_specific(reads1, reads2)
    tab = count.count_k1mers_dict([r for f in dbl for r in f], K + 1)
    if kmer_hard_cutoff > 1:
        tab = {k: c for k, c in tab.items() if c >= kmer_hard_cutoff}
    res = extension.run_correction([(k, tab[k]) for k in sorted(tab, reverse=True)], min_weight=min_weight, min_length=min_length,
                                   comp_size_threshold=partition_size)
    pv = part_vectors or []
    nc, k2c = partition.build_partitions([b[0] for b in res.big_components], [p[0] for p in pv], [p[1] for p in pv] if pv else None,
                                         res.remaining, res.allowed, K)
    if paired:
        o1, o2 = partition.route_reads_paired(dbl[0], dbl[1], nc, k2c, K)
    else:
        o1 = partition.route_reads(dbl[0], nc, k2c, K)
    files, _ = partition.partition_k1mers(nc, k2c, K)
    lines = []
    for i, c in enumerate(res.single_contigs):
        lines += [">Single_%d\n" % i, c + "\n"]
    parts = {}
    for name in nc:
        reads = [o1[name], o2[name]] if paired else [o1[name]]
        g, singles, comps = mbgraph.run_partition(files[name], reads, K, paired)
        sname = "%s_%s" % (sample, name)
        txt = ""
        for c, comp in enumerate(comps):
            tr = sparse_flow.sparse_flow_component(comp["nodes"], comp["edges"], comp["paths"], seed=seed, comp_id=c)
            txt += sparse_flow.fasta_records(sname, str(c), tr)
        txt += sparse_flow.single_nodes_fasta(sname, singles)
        parts[name] = {"reconstructed_fasta": txt, "graph": mbgraph.canonical(singles, comps)}
        lines += txt.splitlines(True)
    return {"partitions": parts, "all_reconstructed": lines, "final": post.finalize(lines, double_stranded),
            "contigs": res.contigs, "n_k1mers": len(tab)}
