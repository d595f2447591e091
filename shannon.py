#!/usr/bin/env python3
"""shannon.py -- command line of the MI355X-native Shannon hot path.

Keeps the reference CLI (sreeramkannan/Shannon shannon.py:145-321) for the flags that drive the
hot path and produces the same products: OUT/shannon.fasta, OUT/log.txt, OUT/TEMP/ (shannon.py:
634-638).  Flags that only select external tools outside the path (quorum, kallisto, --compare,
--filter_FP) are accepted and reported as not built.

    python shannon.py -o OUT --single reads.fasta            [-K 25] [--partition 500]
    python shannon.py -o OUT --left r1.fasta --right r2.fasta [-s / --ss / --strand_specific]
    ... [--kmer_hard_cutoff N]   k1-mers counted fewer than N times are dropped (`jellyfish dump -L N`, shannon.py:237-241, 441; default 1)
    ... [--kmer_soft_cutoff N]   hyp_min_weight: seed threshold + hyperbola of the contig stage (shannon.py:243-247, 457; default 3)
    python shannon.py -o OUT --left r1.fasta --right r2.fasta -p 8        # one rank per GPU (the reference's -p nJobs, shannon.py:527-566)

-p N / --gpus N: the reference fans its partitions out over nJobs processes (GNU parallel, shannon.py:527-566); here the N jobs
are N ranks, one per GPU of the node (torch.distributed over RCCL): this process -- which has made no GPU call -- starts them as
children (torch.distributed.run on 127.0.0.1) and returns their exit code; every rank ingests its slice of the reads, the k1-mer
buckets are exchanged once, partitions are dealt to the ranks, rank 0 merges and writes OUT/ (shannon_amd/distributed.py).
-p is capped at the number of GPUs the node shows (one GPU: the one-process path, partitions concurrently on host threads).
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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, args):
    """start the n ranks as fresh child processes (torch.distributed.run, rendezvous on 127.0.0.1) and hand their exit code on;
    this parent makes no GPU call"""
    import subprocess
    env = dict(os.environ, SHN_CLI_RANKS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(args)
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def rank_main(out_dir, reads, K, partition_size, min_weight, min_length, double_stranded, ignored, noted, kmer_hard_cutoff=1):
    """one rank of an N-rank run: its slice of the reads (by index, contiguous), shannon_amd.distributed.assemble_distributed,
    rank 0 writes OUT/ (shannon.fasta, log.txt, TEMP/<sample>_allalgo_output/all_reconstructed.fasta and the contig files; the
    per-partition graph files stay with the ranks that owned the partitions)"""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("SHN_CLI_BACKEND") == "gloo"
    if os.environ.get("SHN_CLI_LAUNCH_PROBE"):
        # CPU test of the launch (tests/test_cli_launcher.py): the ranks meet over gloo, rank 0 says what the launch resolved to;
        # nothing touches the GPU
        dist.init_process_group("gloo")
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        if rank == 0:
            print("launch probe: %d ranks met, K=%d, partition=%d, double_stranded=%s, reads=%s, out=%s, min_weight=%d, kmer_hard_cutoff=%d"
                  % (int(one.item()), K, partition_size, double_stranded, ",".join(os.path.basename(p) for p in reads), out_dir, min_weight,
                     kmer_hard_cutoff))
        dist.destroy_process_group()
        return 0
    dev_index = 0 if share else local
    torch.cuda.set_device(dev_index)
    if share:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    import shannon_amd
    if os.environ.get("SHN_MALLOC_TUNE", "1") != "0":
        shannon_amd.malloc_tune()
    from shannon_amd import device, distributed, exchange, kmers_for_component as kfc, _lib
    sample = os.path.basename(os.path.normpath(out_dir))
    temp = os.path.join(out_dir, "TEMP")
    log = None
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        os.makedirs(temp)
        log = open(os.path.join(out_dir, "log.txt"), "w")

    def say(msg):
        if rank == 0:
            line = "%s: %s" % (time.asctime(), msg)
            print(line)
            log.write(line + "\n")
    say("Starting Shannon run (MI355X hot path %s, %d ranks)" % (VERSION, world))
    if ignored:
        say("WARNING: flags outside the hot path ignored: " + " ".join(ignored))
    for msg in noted:
        say("NOTE: " + msg)
    ctx = device.Context(dev_index)
    T = {}
    t0 = time.time()
    paired = len(reads) == 2
    # every rank ingests ITS share of the files (by bytes; distributed.ingest_rank_slice) -- the records [n r / W, n (r + 1) / W) of the
    # job, in the files' order; files that cannot be shared out that way (.gz, reads of several lengths, multi-line FASTA) are read
    # whole on every rank as before
    ing_stats = {}
    sl = distributed.ingest_rank_slice(reads, rank, world, None, exchange.coll_device(torch.device("cuda", dev_index), None), stats=ing_stats)
    mats = []
    if sl is not None:
        q1 = sl[0][0]
        q2 = sl[0][1] if paired else None
        n = sl[1]
        lo, hi = rank * n // world, (rank + 1) * n // world
    for p in (reads if sl is None else []):
        try:
            _d, r = device.Reads.ingest(None, p)                # (the host code matrix only: the rank uploads its own slice below)
            if isinstance(r, device.RaggedCodes):
                raise _lib.ShannonError("unsupported: ragged reads on the N-rank path")
        except _lib.ShannonError as ex:
            if "unsupported" not in str(ex):
                raise
            seqs = read_fasta(p)
            L = len(seqs[0]) if seqs else 0
            if any(len(x) != L for x in seqs):
                say("ERROR: the N-rank path takes reads of one length; run without -p / --gpus")
                return 2
            code = np.full(256, 4, np.uint8)
            for j, c in enumerate(b"ACGT"):
                code[c] = j
            r = code[np.frombuffer("".join(seqs).encode(), dtype=np.uint8)].reshape(len(seqs), L) if seqs else np.zeros((0, 1), np.uint8)
        mats.append(r)
    if sl is None:
        if paired and len(mats[0]) != len(mats[1]):
            say("ERROR: --left and --right hold different numbers of reads")
            return 2
        n = len(mats[0])
        lo, hi = rank * n // world, (rank + 1) * n // world
        q1 = np.ascontiguousarray(mats[0][lo:hi])
        q2 = np.ascontiguousarray(mats[1][lo:hi]) if paired else None
        del mats
    T["ingest path"] = "byte share of the files" if sl is not None else "whole files on every rank"
    if ing_stats:
        T["ingest bytes scanned by this rank"] = ing_stats["bytes_scanned"]
        T["ingest bytes of the files"] = ing_stats["file_bytes"]
    d1 = device.Reads.from_codes(ctx, q1)
    d2 = device.Reads.from_codes(ctx, q2) if paired else None
    T["ingest"] = time.time() - t0
    say("Processed No of reads:%d, Avg. Read length: %.2f (every rank holds a slice of %d of them)" % (n, q1.shape[1] if n else 0, hi - lo))
    ops = distributed.GpuOps(ctx, d1, d2, kfc.ReadStore(q1, q2), K)
    res = distributed.assemble_distributed(ops, K, partition_size, sample, 0, timings=T, double_stranded=double_stranded,
                                           min_weight=min_weight, min_length=min_length, kmer_hard_cutoff=kmer_hard_cutoff)
    rc = 0
    if rank == 0:
        say("%d K-mers loaded; %d contigs; %d partitions" % (res["n_k1mers"], len(res["contigs"]), len(res["partitions"])))
        ai = os.path.join(temp, sample + "_algo_input")
        os.makedirs(ai)
        with open(os.path.join(ai, "k1mer.dict_contig"), "w") as f:
            f.write("".join(c + "\n" for c in res["contigs"]))
        alld = os.path.join(temp, sample + "_allalgo_output")
        os.makedirs(alld)
        with open(os.path.join(alld, "all_reconstructed.fasta"), "w") as f:
            for name in res["partitions"]:
                f.write(res["partitions"][name])
        final = res["final"]
        if hasattr(final, "fasta"):
            with open(os.path.join(out_dir, "shannon.fasta"), "wb") as f:
                f.write(final.fasta())
        else:
            with open(os.path.join(out_dir, "shannon.fasta"), "w") as f:
                for name, seq in final.items():
                    f.write(">%s\n%s\n" % (name, seq))
        say("All partitions completed: %d transcripts reconstructed" % len(final))
        say("stage seconds (rank 0): " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in T.items()}))
        log.close()
        print("-------------------------------------------------")
        print(time.asctime() + ": Shannon Run Completed")
        print("-------------------------------------------------")
    dist.barrier()
    d1.close()
    if d2 is not None:
        d2.close()
    ctx.close()
    dist.destroy_process_group()
    return rc


def main(argv):
    K, partition_size, nJobs = 24, 500, 1                     # shannon.py:58,65,67
    out_dir, reads, double_stranded = None, [], True
    min_weight, min_length = 3, 75                            # hyp_min_weight, hyp_min_length: shannon.py:56-57
    kmer_hard_cutoff = 1                                      # jellyfish_kmer_cutoff: shannon.py:55
    i = 1
    ignored, noted = [], []
    takes_value = ("-o", "--single", "--left", "--right", "-K", "-p", "--gpus", "--partition", "--kmer_hard_cutoff", "--kmer_soft_cutoff")
    n_gpus = 0
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
        if a == "--gpus":
            n_gpus = int(argv[i + 1]); i += 2; continue
        if a == "--partition":
            partition_size = int(argv[i + 1]); i += 2; continue
        if a == "--kmer_hard_cutoff":
            # shannon.py:237-241, 441: `jellyfish dump -L N` -- k1-mers counted fewer than N times never enter k1mer.dict_org
            kmer_hard_cutoff = int(argv[i + 1]); i += 2
            print("OPTIONS --kmer_hard_cutoff: Kmer hard cutoff set to " + str(kmer_hard_cutoff)); continue
        if a == "--kmer_soft_cutoff":
            # shannon.py:243-247, 457: hyp_min_weight -> run_correction's min_weight (the seed threshold, extension_correction.py:345,
            # and the hyperbola of the accept filter, :361)
            min_weight = int(argv[i + 1]); i += 2
            print("OPTIONS --kmer_soft_cutoff: Kmer soft cutoff set to " + str(min_weight)); continue
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
    # ---- ranks (shannon.py:527-566: the reference's nJobs processes).  Decided and started before anything touches the GPU:
    # torch.cuda.device_count() does not initialise it, and a process that has must never be replaced or forked into ranks.
    in_rank = "WORLD_SIZE" in os.environ and "RANK" in os.environ and os.environ.get("SHN_CLI_RANKS") == "1"
    if not in_rank:
        want = n_gpus if n_gpus > 0 else nJobs
        if want > 1:
            import torch
            have = torch.cuda.device_count()
            share = os.environ.get("SHN_CLI_BACKEND") == "gloo"          # (development / tests: several ranks on ONE GPU, collectives over gloo)
            ranks = want if share else min(want, max(1, have))
            if ranks > 1:
                return launch_ranks(ranks, argv[1:])
            if n_gpus > 1:
                print("NOTE: --gpus %d asked for, the node shows %d GPU(s): one process" % (n_gpus, have))
    if in_rank:
        return rank_main(out_dir, reads, K, partition_size, min_weight, min_length, double_stranded, ignored, noted, kmer_hard_cutoff)
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
        noted.append("-p %d: one GPU in use -- the partitions run concurrently on host threads and the GPU inside this process (with several "
                     "GPUs, -p N / --gpus N is N ranks)" % nJobs)
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
                                       double_stranded=double_stranded, timings=T, kmer_hard_cutoff=kmer_hard_cutoff)
    else:
        R = pipeline.assemble(ctx, r[0], r[1] if paired else None, K=K, partition_size=partition_size, min_weight=min_weight,
                              min_length=min_length, sample=sample, seed=0, double_stranded=double_stranded, timings=T,
                              kmer_hard_cutoff=kmer_hard_cutoff)
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
