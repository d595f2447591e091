#!/usr/bin/env python3
"""shannon.py -- command line of the MI355X-native Shannon hot path.

Keeps the reference CLI (sreeramkannan/Shannon shannon.py:145-321) for the flags that drive the
hot path and produces the same products: OUT/shannon.fasta, OUT/log.txt, OUT/TEMP/ (shannon.py:
634-638).  Flags that only select external tools outside the path (quorum, kallisto, --compare,
--filter_FP) are accepted and reported as not built.

    python shannon.py -o OUT --single reads.fasta            [-K 25] [--partition 500]
    python shannon.py -o OUT --left r1.fasta --right r2.fasta [-s / --ss / --strand_specific]
"""
import os, sys, time, json

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
VERSION = "0.1.0-mi355x"


def usage():
    print(__doc__)


def read_fasta(path):
    """2-line or multi-line FASTA/FASTQ (plain or .gz) -> list of sequences (upper case kept as in the file)."""
    import gzip
    seqs = []
    with (gzip.open(path, "rt") if path.endswith(".gz") else open(path)) as f:
        first = f.readline()
        f.seek(0)
        if first.startswith("@"):
            lines = f.read().splitlines()
            return [lines[i + 1].strip() for i in range(0, len(lines) - 1, 4)]
        cur = None
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


def main(argv):
    K, partition_size, nJobs = 24, 500, 1                     # shannon.py:58,65,67
    out_dir, reads, double_stranded = None, [], True
    min_weight, min_length = 3, 75                            # shannon.py:55-56
    i = 1
    ignored, noted = [], []
    takes_value = ("-o", "--single", "--left", "--right", "-K", "-p", "--partition", "--kmer_hard_cutoff")
    while i < len(argv):
        a = argv[i]
        if a in takes_value and i + 1 >= len(argv):
            print("ERROR: %s needs a value" % a)
            return 2
        if a in ("--help", "-h"):
            usage(); return 0
        if a == "--version":
            print(VERSION); return 0
        if a == "-o":
            out_dir = argv[i + 1]; i += 2; continue
        if a == "--single":
            reads = [argv[i + 1]]; i += 2; continue
        if a == "--left":
            reads = [argv[i + 1]] + reads[1:]; i += 2; continue
        if a == "--right":
            reads = reads[:1] + [argv[i + 1]]; i += 2; continue
        if a == "-K":
            K = int(argv[i + 1]); i += 2; continue
        if a == "-p":
            nJobs = int(argv[i + 1]); i += 2; continue
        if a == "--partition":
            partition_size = int(argv[i + 1]); i += 2; continue
        if a == "--kmer_hard_cutoff":
            min_weight = int(argv[i + 1]); i += 2; continue
        if a in ("-s", "--ss", "--strand_specific"):
            # shannon.py:166-207, 407-411: no strand doubling; of a pair, RC(reads_2) stands for reads_2
            double_stranded = False; i += 1; continue
        if a in ("--inMem", "--fasta", "--fastq"):
            i += 1; continue
        if a in ("--inDisk", "--only_reads"):
            noted.append("%s: the stages hand their data over in memory (the reference's --inMem contract); TEMP/ holds the per-stage "
                         "products but not reads{comp}.fasta / component*k1mers_allowed.dict (shannon_amd/reference_api.py writes those "
                         "when a single stage is driven through the reference's file interface)" % a)
            i += 1; continue
        if a == "--kmer_soft_cutoff":
            noted.append("--kmer_soft_cutoff %s: not used by the hot path (the reference passes it to jellyfish dump -L only with --filter_FP)"
                         % (argv[i + 1] if i + 1 < len(argv) else "(no value)"))
            i += 2; continue
        if a in ("--compare", "--kallisto_cutoff"):
            ignored.append(a); i += 2; continue
        ignored.append(a); i += 1
    if out_dir is None or not reads:
        print("ERROR: need -o OUT and --single F or --left F1 --right F2")
        print("Try running python shannon.py --help for a short manual")
        return 2
    if os.path.exists(out_dir) and os.listdir(out_dir):
        print("ERROR: output directory is not empty")              # shannon.py:255-257
        return 2
    if K + 1 > 32:
        print("ERROR: K+1 must be <= 32"); return 2
    os.makedirs(out_dir, exist_ok=True)
    sample = os.path.basename(os.path.normpath(out_dir))
    temp = os.path.join(out_dir, "TEMP")
    os.makedirs(temp)
    log = open(os.path.join(out_dir, "log.txt"), "w")

    def say(msg):
        line = "%s: %s" % (time.asctime(), msg)
        print(line)
        log.write(line + "\n")

    import shannon_amd
    if os.environ.get("SHN_MALLOC_TUNE", "1") != "0":
        shannon_amd.malloc_tune()                              # (this program owns its process)
    from shannon_amd import device, pipeline, _lib
    say("Starting Shannon run (MI355X hot path %s)" % VERSION)
    if ignored:
        say("WARNING: flags outside the hot path ignored: " + " ".join(ignored))
    if nJobs != 1:
        noted.append("-p %d: partitions run concurrently on host threads and the GPU inside one process; the value is not used" % nJobs)
    for msg in noted:
        say("NOTE: " + msg)
    ctx = device.Context(0)
    T = {}
    paired = len(reads) == 2
    # the files straight into HBM (shn_reads_ingest / shn_reads_ingest_ragged: 2-line FASTA / 4-line FASTQ, reads of any lengths,
    # bases outside ACGT kept and marked); only what that refuses (multi-line FASTA, malformed records) is read record by record
    import time as _t
    import numpy as np
    t0 = _t.time()
    sets, got = None, []
    try:
        for p in reads:                                        # (a list built step by step: what was ingested before a refusal is closed below)
            got.append(device.Reads.ingest(ctx, p))
        if len(set(len(g[0]) for g in got)) == 1:
            sets, r = [g[0] for g in got], [g[1] for g in got]
            if any(isinstance(x, device.RaggedCodes) for x in r):      # one mate file ragged, the other not: both as flat codes + offsets
                r = [x if isinstance(x, device.RaggedCodes) else
                     device.RaggedCodes(x.reshape(-1), np.arange(x.shape[0] + 1, dtype=np.uint64) * np.uint64(x.shape[1])) for x in r]
    except _lib.ShannonError as ex:
        if "unsupported" not in str(ex):
            raise
    if sets is None:
        for g in got:                                          # ingested but declined (mate files of different sizes): free the device copies
            g[0].close()
        r = [read_fasta(p) for p in reads]
    T["ingest path"] = "device" if sets is not None else "python"
    T["ingest"] = _t.time() - t0
    avg_len = (sum(len(x) for x in r[0]) / max(1, len(r[0]))) if sets is None else (r[0].total_bases / max(1, len(r[0])) if isinstance(r[0], device.RaggedCodes) else r[0].shape[1])
    say("Processed No of reads:%d, Avg. Read length: %.2f (read files through the %s ingest)" % (len(r[0]), avg_len, T["ingest path"]))
    if sets is not None:
        from shannon_amd import kmers_for_component as kfc
        R = pipeline.assemble_resident(ctx, sets[0], sets[1] if paired else None, kfc.ReadStore(r[0], r[1] if paired else None), K=K,
                                       partition_size=partition_size, min_weight=min_weight, min_length=min_length, sample=sample, seed=0,
                                       double_stranded=double_stranded, timings=T)
    else:
        R = pipeline.assemble(ctx, r[0], r[1] if paired else None, K=K, partition_size=partition_size, min_weight=min_weight,
                              min_length=min_length, sample=sample, seed=0, double_stranded=double_stranded, timings=T)
    say("%d K-mers loaded; %d contigs; %d partitions" % (R.n_k1mers, len(R.extension.contigs), len(R.partitions)))
    # TEMP tree: the per-stage products of the reference (shannon.py:496-513, 584-595)
    from shannon_amd import extension_correction as ec, mbgraph
    ai = os.path.join(temp, sample + "_algo_input")
    os.makedirs(ai)
    ec.write_outputs(R.extension, temp, os.path.join(ai, "k1mer.dict"))
    for name, p in R.partitions.items():
        base = os.path.join(temp, "%s_%s" % (sample, name))
        for sub in ("algo_output", "intermediate"):
            os.makedirs(base + sub)
        mbgraph.write_files(p["singles"], p["components"], base + "intermediate")
        open(os.path.join(base + "algo_output", "reconstructed.fasta"), "w").write(p["reconstructed_fasta"])
        say("%s has completed: %d transcripts" % (base, p["reconstructed_fasta"].count(">")))
    alld = os.path.join(temp, sample + "_allalgo_output")
    os.makedirs(alld)
    open(os.path.join(alld, "all_reconstructed.fasta"), "w").write("".join(R.all_reconstructed))
    if hasattr(R.final, "fasta"):                                   # (the native merge's buffers: the file's text without a string per record)
        with open(os.path.join(out_dir, "shannon.fasta"), "wb") as f:
            f.write(R.final.fasta())
    else:
        with open(os.path.join(out_dir, "shannon.fasta"), "w") as f:
            for name, seq in R.final.items():
                f.write(">%s\n%s\n" % (name, seq))
    say("All partitions completed: %d transcripts reconstructed" % len(R.final))
    say("stage seconds: " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in T.items()}))
    log.close()
    ctx.close()
    print("-------------------------------------------------")
    print(time.asctime() + ": Shannon Run Completed")
    print("-------------------------------------------------")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
