#!/usr/bin/env python3
"""bench.py -- reads/s through the Shannon hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch of synthetic 2x100 bp reads already resident
(2-bit packed) in HBM.  N=1 workload = BASELINE configs[2], the largest single-GPU configuration:
100M reads (50M pairs), k=25 (-K 25, k1=26), 20,000 genes, multi-component, --partition 500
(`--config 1` = configs[1]: 10M reads of one gene family; `--config 4s` = the one-GPU slice of
configs[4]: 100M reads, -K 31, exons up to 5 kb).  N>1 (default `--scaling strong`): the SAME
configs[2] batch hash-sharded over the N ranks = BASELINE configs[3] -- every rank generates its
own contiguous slice of the batch's 250,000-pair chunks (a chunk's reads depend only on the seed and
the chunk's index, so the union over the ranks is the N=1 batch and the transcripts' digest is the
N=1 digest), the ranks exchange their (key,count) buckets with one all-to-all over RCCL/xGMI.
`--scaling weak`: N gene families of the configs[1] kind, 10M reads per rank.
`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (fresh
child processes through torch.distributed.run, from a parent that has made no GPU call) and relays
rank 0's JSON line.  Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import shannon_amd
if os.environ.get("SHN_MALLOC_TUNE", "1") != "0":
    shannon_amd.malloc_tune()           # this program owns its process: heap-served blocks, no trimming (before anything allocates much)
import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over this same command at the
# default workload; tools/summarize_prof.py -> profiles/r01_traffic.json): 2 x FETCH_SIZE (gfx950 tallies 128-B
# requests at 64 B, MI355X_MICROARCH.md HBM; profiles/r03_fetch_calibration.txt: on this chip EVERY read request the L2 sends to
# memory is a 128-byte one, random 8-byte gathers included) + WRITE_SIZE.  Valid for the named workload only.
TRAFFIC = {}
def _newest_traffic(cfg):
    """the newest committed profiles/rNN_traffic_config<cfg>.json (r01_traffic.json for configs[1])"""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic_config%s.json" % cfg)))
    if cfg == "1":
        names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json"))) + names
    return names[-1] if names else None


for _cfg in ("1", "2", "4s", "2p"):
    _fn = _newest_traffic(_cfg)
    try:
        TRAFFIC["config%s" % _cfg] = json.load(open(_fn))["traffic_bytes_per_launch"]
        TRAFFIC["config%s_file" % _cfg] = os.path.basename(_fn)
    except (TypeError, OSError, ValueError, KeyError):
        sys.stderr.write("[bench] no PMC traffic file for config %s under profiles/: the kernel table carries no measured bytes for it\n" % _cfg)


CHUNK_PAIRS = 250_000          # the batch is generated in chunks of this many pairs; a chunk depends on (seed, chunk index) only


def chunk_range(total_pairs, world, rank):
    """(first chunk, pairs) of the contiguous slice of the batch's chunks that rank `rank` of `world` generates and holds"""
    n_chunks = (total_pairs + CHUNK_PAIRS - 1) // CHUNK_PAIRS
    if n_chunks < world:                                       # (a rank without a chunk would run the path on an empty batch)
        raise SystemExit("bench.py: --reads %d is %d chunk(s) of %d pairs: fewer than the %d ranks of the job -- use at least %d reads "
                         "or fewer ranks" % (2 * total_pairs, n_chunks, CHUNK_PAIRS, world, 2 * world * CHUNK_PAIRS))
    lo, hi = rank * n_chunks // world, (rank + 1) * n_chunks // world
    return lo, min(total_pairs, hi * CHUNK_PAIRS) - min(total_pairs, lo * CHUNK_PAIRS)


def _chunk_seed(read_seed, chunk):
    z = (read_seed * 0x9E3779B97F4A7C15 + (chunk + 1) * 0xD1B54A32D192ED03) & ((1 << 64) - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
    return (z ^ (z >> 31)) & ((1 << 62) - 1)


def gen_reads(n_pairs, seed, n_genes, device, read_seed=None, families=0, first_chunk=0, exon_len=(80, 600), chain_exons=0):
    """synthetic pairs on the GPU (torch RNG; same model as shannon_amd/synth.py), generated in chunks of CHUNK_PAIRS pairs:
    chunk c (counted from first_chunk) is drawn from a generator seeded with (read_seed, c) alone, so any rank can produce any
    slice of the batch and the union of the ranks' slices is the one-GPU batch.
    families > 0: that many gene families of the configs[1] kind (rich in isoforms) instead of n_genes plain genes."""
    from shannon_amd import synth
    # config 2 ("single component"): one gene family rich enough to give a multi-contig component
    if chain_exons:
        # --config 2p: genes linked by shared repeats (synth.make_repeat_family): one component of the contig graph per strand with about
        # two contigs per gene -- the input that takes the gpmetis branch (kmers_for_component.py:207-237)
        # (a repeat element per 5 genes, every fourth gene of a repeat's class carrying the next one as well: the repeats chain into one
        # component while no graph node gets more than a handful of branches -- three repeats for 30,000 genes would put a 10,000 x 10,000
        # X-node into one partition, and 20 genes per repeat still cost the graph stage and the sparse flow 23 s per step)
        iso = synth.make_repeat_family(chain_exons, seed, n_repeats=max(3, chain_exons // 5))
        lens = np.array([len(t) for t in iso], dtype=np.int64)
        wts = np.random.Generator(np.random.PCG64(seed + 1)).lognormal(0.0, 0.5, size=len(iso)) * (lens - 300 + 1)
        wts /= wts.sum()
    elif families:
        # every family is built like the configs[1] one (family 0 IS the configs[1] gene) and gets the same share of reads
        iso, wl = [], []
        for f in range(families):
            fi, _ = synth.make_transcriptome(1, seed + 7919 * f, n_isoforms=(4, 6), n_exons=(8, 12))
            fl = np.array([len(t) for t in fi], dtype=np.int64)
            w = np.random.Generator(np.random.PCG64(seed + 1 + 7919 * f)).lognormal(0.0, 1.5, size=len(fi)) * (fl - 300 + 1)
            iso += fi
            wl.append(w / w.sum() / families)
        lens = np.array([len(t) for t in iso], dtype=np.int64)
        wts = np.concatenate(wl)
    else:
        kw = dict(n_isoforms=(4, 6), n_exons=(8, 12)) if n_genes == 1 else {}
        iso, _ = synth.make_transcriptome(n_genes, seed, exon_len=exon_len, **kw)
        lens = np.array([len(t) for t in iso], dtype=np.int64)
        rng = np.random.Generator(np.random.PCG64(seed + 1))
        expr = rng.lognormal(0.0, 1.5, size=len(iso))
        wts = expr * (lens - 300 + 1)
        wts /= wts.sum()
    read_seed = seed + 2 if read_seed is None else read_seed
    g = torch.Generator(device=device)
    cat = torch.as_tensor(np.concatenate(iso), device=device)
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), device=device)
    tl = torch.as_tensor(lens, device=device)
    out1 = np.empty((n_pairs, 100), dtype=np.uint8)
    out2 = np.empty((n_pairs, 100), dtype=np.uint8)
    ar = torch.arange(100, device=device)
    # isoform of every fragment by inverse CDF (float64): reproducible run to run (torch.multinomial on the device was not --
    # the same seed gave tables of 724,313,063 or 724,313,437 distinct k1-mers on different boxes)
    cdf = torch.as_tensor(np.cumsum(wts / wts.sum()), device=device, dtype=torch.float64)
    for c, s in enumerate(range(0, n_pairs, CHUNK_PAIRS)):
        n = min(CHUNK_PAIRS, n_pairs - s)
        g.manual_seed(_chunk_seed(read_seed, first_chunk + c))
        iso_i = torch.searchsorted(cdf, torch.rand(n, device=device, generator=g, dtype=torch.float64)).clamp_(max=len(lens) - 1)
        start = (torch.rand(n, device=device, generator=g, dtype=torch.float64) * (tl[iso_i] - 299)).long() + offs[iso_i]
        a = cat[start[:, None] + ar]
        b = 3 - cat[(start + 299)[:, None] - ar]
        for m in (a, b):
            e = torch.rand(m.shape, device=device, generator=g) < 0.005
            sub = torch.randint(1, 4, m.shape, device=device, generator=g, dtype=torch.uint8)
            m[e] = (m[e] + sub[e]) & 3
        out1[s:s + n] = a.cpu().numpy()
        out2[s:s + n] = b.cpu().numpy()
    return out1, out2


def _final_sha(final):
    """order-free digest of the final transcript set (run-to-run determinism is checked with it)"""
    import hashlib
    h = hashlib.sha256()
    for sq in sorted(final.values()):
        h.update(sq.encode() + b"\n")
    return h.hexdigest()[:16]


def ingest_rate(ctx, r1, r2, n_pairs):
    """SURVEY 8 row f2 beside the resident number: the first n_pairs pairs of the batch written as two 2-line FASTA texts in host
    memory (what the reference is handed as files, here already in the page cache), timed through shn_reads_ingest (host-thread
    parse -> pinned double buffers -> H2D -> pack_kernel, host code matrix included).  Returns the dict of the JSON line."""
    from shannon_amd import device
    A = np.frombuffer(b"ACGT", np.uint8)

    def fasta(codes):
        n, L = codes.shape
        w = 1 + 9 + 1                                         # ">%09d\n"
        rec = np.empty((n, w + L + 1), np.uint8)
        rec[:, 0] = ord(">")
        idx = np.arange(n, dtype=np.int64)
        for d in range(9):
            rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        rec[:, 10] = 10
        rec[:, w:w + L] = A[codes]
        rec[:, w + L] = 10
        return rec.reshape(-1)
    texts = [fasta(r1[:n_pairs]), fasta(r2[:n_pairs])]
    device.Reads.ingest(ctx, texts[0])[0].close()                          # first-use costs (pinning the staging pair, kernels) outside
    t = time.time()
    got = [device.Reads.ingest(ctx, x) for x in texts]
    dt = time.time() - t
    ok = all(np.array_equal(g[1][::997], c[:n_pairs][::997]) for g, c in zip(got, (r1, r2)))
    for g in got:
        g[0].close()
    nb = sum(len(x) for x in texts)
    return {"reads_per_s": 2 * n_pairs / dt, "text_GB_per_s": nb / dt / 1e9, "seconds": dt, "host_cpus": _host_cpus(), "matrix_ok": bool(ok),
            "sample": "first %d pairs of the batch as two 2-line FASTA texts in host memory (%.2f GB), shn_reads_ingest incl. the host code matrix" % (n_pairs, nb / 1e9)}


def native_graph_baseline(R, store, K):
    """one partition of the run (the one of median size) through the native graph stage WITHOUT the device --
    shn_mbgraph_run with ctx = NULL: K-mer graph built and condensed sequentially, seeds matched on the host -- i.e. the reference's
    multibridging.main for that partition restated in C++ on this box's host cores."""
    from shannon_amd import mbgraph_native, kmers_for_component as kfc
    P = R.partitioning
    names = sorted(P["routes"], key=lambda nm: len(P["routes"][nm]))
    nm = names[len(names) // 2]
    contigs = P["new_components"][nm]
    rows = kfc._rows_bytes(contigs, K + 1)
    n_rows = len(rows) // (K + 1)
    idx = P["routes"][nm]
    b1, o1, rc1, enc = store.gather_codes(idx, 1)
    b2, o2, rc2 = b1, o1, (1 - rc1).astype(np.uint8)
    t = time.time()
    g = mbgraph_native.run_partition_handle(rows, n_rows, K, b1, o1, b2, o2, ctx=None, enc=enc, rc1=rc1, rc2=rc2)
    dt = time.time() - t
    g.close()
    return {"partition": nm, "contigs": len(contigs), "k1mers": n_rows, "pairs_routed": int(len(idx)), "seconds": dt, "pairs_per_s": len(idx) / dt,
            "note": "shn_mbgraph_run(ctx=NULL) on the host only; the whole run has %d partitions and %d routed pairs" % (len(names), sum(len(v) for v in P["routes"].values()))}


def overlapped_steps(ctx, sets, store, K, n, want_sha, n_reads):
    """Two batches in flight: count -> extension -> partition / route -> unitigs of batch i+1 on the main thread and context while
    graph -> sparse flow -> merge of batch i run on a second thread and context (the GPU is idle for most of the second half of a
    step, the host for most of the first).  A throughput figure for a stream of samples -- reported beside `value`, which stays
    one batch at a time.  Every batch's transcripts must equal the sequential run's.
    SHN_OVERLAP_SPLIT=early: the cut behind the extension -- partition / route / unitigs with the back half, whose context then has
    stage workspaces of its own (shn_ctx_own_workspaces).  Measured in round 6: the same figure (1.34 s either way at configs[2]: the
    stages moved over are GPU work too, and the chip is what the halves share: route 0.09 -> 0.39 s, unitigs 0.05 -> 0.17 s beside an
    extension)."""
    from concurrent.futures import ThreadPoolExecutor
    from shannon_amd import device, pipeline
    # the second context works on a stream of its OWN (round 6).  Until then both contexts were made on the legacy NULL stream, which
    # orders everything issued on it: whatever the back half put there itself (the merge's uploads and fingerprint passes, its
    # synchronisations) waited behind the front half's whole queue.  SHN_OVERLAP_STREAM=0: the NULL stream again; 2: a high-priority stream.
    mode = os.environ.get("SHN_OVERLAP_STREAM", "2")
    own = None if mode == "0" else torch.cuda.Stream(device=torch.device("cuda", 0), priority=(-1 if mode == "2" else 0))
    split = os.environ.get("SHN_OVERLAP_SPLIT", "late")
    ctx_b = device.Context(0, stream=own.cuda_stream if own is not None else None, own_workspaces=(split == "early"))
    pool = ThreadPoolExecutor(max_workers=1)
    try:
        def front():
            return pipeline.assemble_resident(ctx, sets[0], sets[1], store, K=K, sample="bench", seed=1, timings={}, defer_back=("early" if split == "early" else True))
        prev = pool.submit(front(), ctx_b)                     # one batch ahead, untimed
        torch.cuda.synchronize()
        t = time.time()
        finals, stage_s = [], []
        for _ in range(n):
            fin = front()
            res = prev.result()
            finals.append(res.final)
            stage_s.append(dict(res.timings))
            prev = pool.submit(fin, ctx_b)
        last = prev.result().final                             # (the batch started before the clock ends inside it: n batches in, n out)
        ctx.sync(); ctx_b.sync()
        torch.cuda.synchronize()
        dt = time.time() - t
        finals.append(last)
        ok = all(_final_sha(f) == want_sha for f in finals)
        # every stage's wall time with the other half running beside it (the batches whose back half ran inside the clock, the
        # first one -- started beside nothing -- left out): set against config.host_stage_seconds_per_step this names what the
        # overlap costs each stage
        per = stage_s[1:] or stage_s
        stages = {k: sum(t.get(k, 0.0) for t in per) / len(per) for k in (per[0] if per else {})}
        return {"value": n_reads * n / dt, "unit": "reads/s", "steps": n, "ms_per_step": 1000.0 * dt / n, "transcripts_equal_sequential_run": bool(ok),
                "stage_seconds_per_step_side_by_side": stages,
                "back_half_stream": {"0": "legacy NULL stream", "1": "own stream", "2": "own high-priority stream"}.get(mode, mode),
                "cut": "behind the extension (the back half's context has stage workspaces of its own)" if split == "early" else "behind the unitig batch",
                "note": "two batches in flight (second half of batch i beside the first half of batch i+1, two contexts); not the headline value"}
    except Exception as ex:                                    # an extra: its failure must not cost the bench line
        return {"error": str(ex)[:300]}
    finally:
        pool.shutdown(wait=True)
        ctx_b.close()


def _host_cpus():
    from shannon_amd import _lib
    return _lib.host_cpus()


def native_front_baseline(k1, r1, r2, n_pairs):
    """the two stages that dominate the reference's run time -- counting (its Jellyfish) and the greedy extension (its
    Python loop) -- plus the accept filter / duplicate_check / contig graph, natively on ONE host core over the first n_pairs pairs
    of the batch: oracle/count_c.c, oracle/ext_c.c (C restatements, pinned in tests/test_oracle_c.py) and the sequential contig
    stage of the product's native host code (shn_contig_graph: the reference's loop in C++)."""
    import math
    from oracle import build_c
    from shannon_amd import extension_correction as ec
    codes = np.concatenate([r1[:n_pairs], r2[:n_pairs]])
    t = time.time()
    keys, cnts, nw = build_c.count_canonical(codes, k1, True)
    t_count = time.time() - t
    t = time.time()
    seed, nr, nl, tw, bases = build_c.extend(keys, cnts, k1, 3, strings=False)
    t_ext = time.time() - t
    # accept filter (extension_correction.py:361) on the arrays, then the candidates' strings for the native contig stage
    L = k1 + nr.astype(np.int64) + nl.astype(np.int64)
    nk = (nr.astype(np.int64) + nl.astype(np.int64) + 1)
    sure = (L >= 75) & (L * np.power(tw.astype(np.float64) / np.maximum(1, nk), 0.25) >= 2 * 75 * math.pow(3, 0.25))
    A = np.frombuffer(b"ACGT", np.uint8)
    off = np.concatenate([[0], np.cumsum(nr.astype(np.int64) + nl.astype(np.int64))])
    cands = []
    for i in np.nonzero(sure)[0].tolist():
        s = int(seed[i])
        a, b, at = int(nr[i]), int(nl[i]), int(off[i])
        cands.append(A[bases[at + a:at + a + b][::-1]].tobytes().decode() + "".join("ACGT"[(s >> (2 * (k1 - 1 - j))) & 3] for j in range(k1)) +
                     A[bases[at:at + a]].tobytes().decode())
    t = time.time()
    acc = ec.contig_stage(cands, k1)[0] if cands else np.zeros(0, np.int32)
    t_contig = time.time() - t
    tot = t_count + t_ext + t_contig
    return {"value": 2 * n_pairs / tot, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the batch: count (oracle/count_c.c) %.2f s, greedy extension (oracle/ext_c.c) %.2f s over %d distinct "
                      "canonical k1-mers -> %d walks, accept filter + duplicate_check + contig graph (shn_contig_graph, host C++) %.2f s -> %d "
                      "contigs; the graph / sparse-flow / merge stages are not in this figure (graph_stage_native_host_only has one partition)"
                      % (2 * n_pairs, t_count, t_ext, len(keys), len(seed), t_contig, int((np.asarray(acc) > 0).sum())),
            "seconds": {"count": t_count, "extension": t_ext, "contig stage": t_contig}}


def cpu_whole_path_native(K, r1, r2, n_pairs):
    """SURVEY 8d's CPU baseline: the WHOLE path a1-a31 on this box's host cores over the first n_pairs pairs of the batch, native
    wherever a native host form exists, no device anywhere:
      count          oracle/count_c.c (radix sort + run-length count; the reference's Jellyfish)                 one thread | T slices
      extension      oracle/ext_c.c (the greedy walks in seed order; the reference's Python loop)                 one thread
      contig stage   accept filter (numpy), duplicate_check + contig graph + components in host C++ (shn_cgraph,
                     shn_contig_components: the reference's sequential loops)                                      one thread
      partitions     oracle/partition.py (partitions' contigs), reads routed by its numpy form route_pairs_matrix one thread
      graph          shn_mbgraph_run(ctx = NULL) per partition: K-mer graph built and condensed sequentially, seeds matched on the
                     host -- the reference's multibridging.main restated in C++                                   one thread | T threads
      sparse flow    oracle/sparse_flow.py + oracle/lp.py over the exported tables (Python: the LP trials have no host form
                     outside the oracle)                                                                            one thread
      merge          shn_post_finalize_bufs (host C++)                                                             host threads
    Returns the one-thread figure as `value` and the figure with the two stages that parallelise run on T threads."""
    from concurrent.futures import ThreadPoolExecutor
    import math
    from oracle import build_c, partition as opart, sparse_flow as osf
    from shannon_amd import _lib, mbgraph_native, post, kmers_for_component as kfc, extension_correction as ec
    k1 = K + 1
    a1, a2 = np.ascontiguousarray(r1[:n_pairs]), np.ascontiguousarray(r2[:n_pairs])
    sec = {}
    t = time.time()
    codes = np.concatenate([a1, a2])
    keys, cnts, nw = build_c.count_canonical(codes, k1, True)
    sec["count"] = time.time() - t
    t = time.time()
    seed, nr, nl, tw, bases = build_c.extend(keys, cnts, k1, 3, strings=False)
    sec["extension"] = time.time() - t
    t = time.time()
    # accept filter (extension_correction.py:361) on the arrays, the candidates' strings, then duplicate_check + contig graph +
    # components in native host code (shn_cgraph / shn_contig_components: the reference's sequential loops in C++) and the file
    # products of extension_correction.py:458-513 (single contigs, bins of small components, components above --partition)
    Lc = k1 + nr.astype(np.int64) + nl.astype(np.int64)
    nk = nr.astype(np.int64) + nl.astype(np.int64) + 1
    sure = (Lc >= 75) & (Lc * np.power(tw.astype(np.float64) / np.maximum(1, nk), 0.25) >= 2 * 75 * math.pow(3, 0.25))
    A = np.frombuffer(b"ACGT", np.uint8)
    off = np.concatenate([[0], np.cumsum(nr.astype(np.int64) + nl.astype(np.int64))])
    cands = []
    for i in np.nonzero(sure)[0].tolist():
        sd = int(seed[i])
        a_, b_, at = int(nr[i]), int(nl[i]), int(off[i])
        cands.append(A[bases[at + a_:at + a_ + b_][::-1]].tobytes().decode() + "".join("ACGT"[(sd >> (2 * (k1 - 1 - j))) & 3] for j in range(k1)) +
                     A[bases[at:at + a_]].tobytes().decode())
    acc, coff, cnb, cw = ec.contig_stage(cands, k1) if cands else (np.zeros(0, np.int32), [0], [], [])
    contigs = ["buffer"] + [cands[i] for i in np.nonzero(np.asarray(acc))[0].tolist()]
    coff_a, cnb_a, cw_a = np.asarray(coff, dtype=np.uint64), np.asarray(cnb, dtype=np.int32), np.asarray(cw, dtype=np.int32)
    _co, members, comp_off, comp_edges = ec.contig_components(coff_a, cnb_a)
    mem, co = members.tolist(), comp_off.tolist()
    single_contigs, big_components, remaining, cur_size = [], [], [[]], 0
    for j, sz in enumerate(np.diff(comp_off.astype(np.int64)).tolist() if len(comp_off) > 1 else []):
        if sz == 1:
            single_contigs.append(contigs[mem[co[j]]])
        elif sz > 500:
            mm = mem[co[j]:co[j + 1]]
            code_ = {c: i + 1 for i, c in enumerate(mm)}
            lines = ["%d\t%d\t001\n" % (sz, int(comp_edges[j]))]
            o_, nb_, w_ = coff_a.tolist(), cnb_a.tolist(), cw_a.tolist()
            for c in mm:
                lines.append("".join("%d\t%d\t" % (code_[c2], wt) for c2, wt in zip(nb_[o_[c - 1]:o_[c]], w_[o_[c - 1]:o_[c]])) + "\n")
            big_components.append(([contigs[c] for c in mm], "".join(lines)))
        else:
            remaining[-1].extend(contigs[c] for c in mem[co[j]:co[j + 1]])
            cur_size += sz
            if cur_size > 500:
                remaining.append([])
                cur_size = 0
    sec["contig stage"] = time.time() - t
    t = time.time()
    pv = []
    for bc, metis in big_components:
        P = kfc.n_partitions(len(bc), 500)
        p1 = kfc.partition_graph(metis, P, 1000)
        pv.append((p1, kfc.partition_graph(kfc.weight_updated_graph(metis, p1, 5), P, 1000)))
    nc = {}                                          # partition -> contigs, named and ordered as kmers_for_component.py:244-305 names them
    for i, (bc, _m) in enumerate(big_components):
        for j, pid in enumerate(pv[i][0]):
            nc.setdefault("c%d_%s" % (i + 1, pid), []).append(bc[j])
        for j, pid in enumerate(pv[i][1]):
            nc.setdefault("r2_c%d_%s" % (i + 1, pid), []).append(bc[j])
    for i, lst in enumerate(remaining):
        for c in lst:
            nc.setdefault("cremaining%d" % (i + 1), []).append(c)
    routes = opart.route_pairs_matrix(a1, a2, nc, K)
    sec["partition+route"] = time.time() - t
    store = kfc.ReadStore(a1, a2)
    names = [nm for nm in nc]
    code = np.zeros(256, np.uint64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    pw = (np.uint64(4) ** np.arange(k1 - 1, -1, -1, dtype=np.uint64)).astype(np.uint64)

    def graph_of(nm):
        rows = kfc._rows_bytes(nc[nm], k1)
        n_rows = len(rows) // k1
        kk = np.unique(np.concatenate([np.lib.stride_tricks.sliding_window_view(code[np.frombuffer(c.encode(), np.uint8)], K) @ pw[1:] for c in nc[nm] if len(c) >= K])) if nc[nm] else np.zeros(0)
        idx = routes[nm][:10 * len(kk) + 1]                             # multibridging.py:26-30
        b1, o1, rc1, enc = store.gather_codes(idx, 1)
        g = mbgraph_native.run_partition_handle(rows if n_rows else np.zeros(1, np.uint8), n_rows, K, b1, o1, b1, o1, ctx=None, enc=enc, rc1=rc1,
                                                rc2=(1 - rc1).astype(np.uint8) if rc1 is not None else None)
        tabs = g.tables()
        g.close()
        return tabs
    t = time.time()
    tables = [graph_of(nm) for nm in names]
    sec["graph"] = time.time() - t
    T = max(1, min(_lib.host_cpus(), 32))
    t = time.time()
    with ThreadPoolExecutor(max_workers=min(T, max(1, len(names)))) as pool:
        list(pool.map(graph_of, names))
    graph_T = time.time() - t
    t = time.time()
    texts = ["".join(">Single_%d\n%s\n" % (i, c) for i, c in enumerate(single_contigs))]
    for nm, (singles, comps, _log) in zip(names, tables):
        txt = ""
        for c, comp in enumerate(comps):
            txt += osf.fasta_records("cpu_%s" % nm, str(c), osf.sparse_flow_component(comp["nodes"], comp["edges"], comp["paths"], seed=1, comp_id=c))
        texts.append(txt + osf.single_nodes_fasta("cpu_%s" % nm, singles))
    sec["sparse flow"] = time.time() - t
    t = time.time()
    final = post.finalize_texts(texts, True)
    sec["merge"] = time.time() - t
    # counting on T threads: T slices of the sample, one C call each (no merge of the slice tables: an upper bound of what T cores give)
    per = max(1, n_pairs // T)
    slices = [np.concatenate([a1[i * per:(i + 1) * per], a2[i * per:(i + 1) * per]]) for i in range(T) if (i + 1) * per <= n_pairs]
    t = time.time()
    with ThreadPoolExecutor(max_workers=len(slices) or 1) as pool:
        list(pool.map(lambda c: build_c.count_canonical(c, k1, True), slices))
    count_T = time.time() - t
    tot = sum(sec.values())
    tot_T = tot - sec["count"] - sec["graph"] + count_T + graph_T
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    return {"value": 2 * n_pairs / tot, "unit": "reads/s", "cores": 1, "kind": "port", "cpu_model": cpu, "host_threads_available": os.cpu_count(),
            "host_cpus_allowed": _lib.host_cpus(),
            "sample": "first %d reads of the benchmark batch through the whole path a1-a31 on the host, native where a native host form exists "
                      "(oracle/count_c.c, oracle/ext_c.c, shn_cgraph + shn_contig_components, shn_mbgraph_run(ctx=NULL) for all %d partitions, "
                      "shn_post_finalize_bufs; routing in numpy, the sparse flow + LP trials through oracle/*.py): %d distinct k1-mers, %d walks, %d "
                      "contigs, %d routed pairs, %d transcripts, %.1f s on one thread"
                      % (2 * n_pairs, len(names), len(keys), len(seed), len(contigs) - 1, sum(len(v) for v in routes.values()), len(final), tot),
            "seconds": {k: round(v, 3) for k, v in sec.items()},
            "threads": {"value": 2 * n_pairs / tot_T, "unit": "reads/s", "cores": T,
                        "sample": "the same with the two stages that parallelise on %d threads: counting as %d slices (%.2f s, tables unmerged) and the "
                                  "graph stage one partition per thread (%.2f s); the other stages are sequential in the reference too" % (T, len(slices), count_T, graph_T)}}


def cpu_baseline(k1, r1, r2, n_pairs):
    """The CPU restatement of the reference over a bounded sample of the same batch, on this box's host cores (a reported
    baseline, not the target).  `value`: the whole path a1-a31 through oracle/pipeline.py -- pure Python like Shannon itself, one
    host thread.  `count_stage_threads`: the counting stage alone (the reference's Jellyfish, its only multi-threaded stage)
    through the C restatement oracle/count_c.c on T host threads, one slice of reads each.  `contig_stage_native`: the sequential
    duplicate_check + contig_connections loop (extension_correction.py:247-270, 358-397) in C++ on one core over the sample's own
    candidate contigs is too small to time here; see DESIGN.md for its measured cost at full size (25 s)."""
    from oracle import pipeline as opipe, build_c
    from shannon_amd import _lib
    A = np.frombuffer(b"ACGT", np.uint8)
    s1 = [A[r].tobytes().decode() for r in r1[:n_pairs]]
    s2 = [A[r].tobytes().decode() for r in r2[:n_pairs]]
    t = time.time()
    opipe.assemble(s1, s2, K=k1 - 1)
    dt = time.time() - t
    build_c.build()
    codes = np.concatenate([r1[:500000], r2[:500000]])
    t = time.time()
    build_c.count_canonical(codes, k1, True)
    dtc = time.time() - t
    # the same counting on T threads: T slices of 500k reads each (ctypes releases the GIL; the per-slice tables would still
    # have to be merged, which Jellyfish's shared hash avoids -- so this is an upper bound of what T cores give the port)
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(_lib.host_cpus(), 32))
    per = 250000
    slices = [np.concatenate([r1[i * per:(i + 1) * per], r2[i * per:(i + 1) * per]]) for i in range(T) if (i + 1) * per <= len(r1)]
    t = time.time()
    with ThreadPoolExecutor(max_workers=len(slices) or 1) as pool:
        list(pool.map(lambda c: build_c.count_canonical(c, k1, True), slices))
    dtt = time.time() - t
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    return {"value": 2 * n_pairs / dt, "unit": "reads/s", "cores": 1, "kind": "port", "cpu_model": cpu, "host_threads_available": os.cpu_count(), "host_cpus_allowed": _lib.host_cpus(),
            "sample": "first %d reads of the benchmark batch through oracle/pipeline.py (count -> extension -> partition -> graph -> "
                      "sparse flow -> merge), %.1f s" % (2 * n_pairs, dt),
            "count_stage_only": {"value": len(codes) / dtc, "unit": "reads/s", "cores": 1,
                                 "sample": "%d reads, oracle/count_c.c (radix sort + run-length count), %.2f s" % (len(codes), dtc)},
            "count_stage_threads": {"value": sum(len(c) for c in slices) / dtt if slices else None, "unit": "reads/s", "cores": len(slices),
                                    "sample": "%d slices of %d reads, oracle/count_c.c on one host thread each (no merge), %.2f s"
                                              % (len(slices), 2 * per, dtt)}}


PRESETS = {"1": dict(genes=1, reads=10_000_000, K=25, exon_len=(80, 600)),
           "2": dict(genes=20000, reads=100_000_000, K=25, exon_len=(80, 600)),
           # the one-GPU slice of configs[4] (500M reads, k=31, 8 GPUs): a fifth of its reads and of its genes (the per-gene depth
           # of configs[4]), exons up to 5 kb so that the unitigs get long
           "4s": dict(genes=4000, reads=100_000_000, K=31, exon_len=(80, 5000)),
           # not one of BASELINE's configs: the input for row a8 (a component of the contig graph far larger than --partition, cut by
           # the library's partitioner into the reference's 100 parts, twice): 30,000 genes linked by 6,000 shared repeats in a chain, 20 M reads
           "2p": dict(genes=1, reads=20_000_000, K=25, exon_len=(80, 600), chain_exons=30000)}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="", choices=["", "0", "1", "2", "4s", "2p"],
                    help="BASELINE.json configs[i]: 1 = 10M reads / one gene family, 2 = 100M reads / 20,000 genes (on N GPUs with "
                         "--scaling strong: configs[3]), 4s = the one-GPU slice of configs[4]: 100M reads, -K 31, 4,000 genes with exons "
                         "up to 5 kb (default: 2; 1 per rank with --scaling weak on several GPUs)")
    ap.add_argument("--reads", type=int, default=0, help="reads of the batch (2 per pair; per GPU with --scaling weak); default from --config")
    ap.add_argument("--K", type=int, default=0)
    ap.add_argument("--genes", type=int, default=0, help="genes of the synthetic transcriptome; default from --config")
    ap.add_argument("--families", type=int, default=0, help="gene families of the configs[1] kind (default with --scaling weak: one per rank)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="several GPUs: strong (default) = the config's batch is split over the ranks (configs[2] on N GPUs = BASELINE "
                         "configs[3]); weak = every rank gets the config's reads (N gene families of the configs[1] kind by default)")
    ap.add_argument("--overlap-steps", type=int, default=-1,
                    help="one GPU: after the timed steps, that many more with two batches in flight (the host-bound half of a step on a second "
                         "context and thread beside the next batch's GPU-bound half); reported as `overlap`, never as `value` (default: 6 at "
                         "the default workload, 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-distributed", action="store_true", help="run the multi-GPU code path even with one rank")
    return ap


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, json_fd):
    """`python bench.py --gpus N` outside a torch.distributed launch: start the N ranks as fresh child processes (one per GPU,
    torch.distributed.run on 127.0.0.1) from this parent, which has made no GPU call (argparse and subprocess only -- a process
    that has initialised the GPU must never be replaced or forked into ranks), relay rank 0's JSON line, and fail when a child
    fails.  Returns the exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "1")
    import time as _time
    for attempt in (0, 1):
        # (the port is found free and taken by the launcher a moment later: somebody else can take it in between -- a launch that
        # fails within seconds gets one more try on another port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
        t_launch = _time.time()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, cwd=ROOT)
        line = None
        for raw in proc.stdout:                                # the ranks' stdout: the JSON line of rank 0 and nothing else
            txt = raw.decode(errors="replace")
            if txt.lstrip().startswith("{") and '"metric"' in txt:
                line = txt.strip()
            else:
                sys.stderr.write(txt)
        rc = proc.wait()
        if rc == 0 or line is not None or attempt == 1 or _time.time() - t_launch > 15:
            break
        sys.stderr.write("bench.py: the %d-rank launch failed within %.0f s (exit code %d): once more on another port\n" % (n, _time.time() - t_launch, rc))
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank launch failed with exit code %d\n" % (n, rc))
        return rc
    if line is None:
        sys.stderr.write("bench.py: the %d-rank launch printed no JSON line\n" % n)
        return 1
    os.write(json_fd, (line + "\n").encode())
    return 0


def main():
    # stdout carries the JSON line and nothing else: whatever libraries print through C stdio (RCCL's version banner) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    args = build_parser().parse_args()
    if args.config == "0":
        args.config = ""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], json_fd))

    if os.environ.get("SHN_BENCH_WATCHDOG"):          # development aid: dump every thread's stack if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SHN_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    if args.scaling is None:
        args.scaling = "weak" if args.families else "strong"
    if not args.config:
        args.config = "1" if (world > 1 and args.scaling == "weak" and not (args.genes or args.reads)) else "2"
    preset = PRESETS[args.config]
    args.genes = args.genes or preset["genes"]
    args.reads = args.reads or preset["reads"]
    args.K = args.K or preset["K"]
    exon_len = preset["exon_len"]
    chain_exons = preset.get("chain_exons", 0) if not args.families else 0
    is_config = (args.genes, args.reads, args.K) == (preset["genes"], preset["reads"], preset["K"]) and not args.families
    total_reads = args.reads                                   # of the whole job's batch (strong) / per rank (weak)
    first_chunk = 0
    if args.scaling == "strong" and world > 1:
        first_chunk, pairs = chunk_range(args.reads // 2, world, rank)
        args.reads = 2 * pairs                                 # this rank's slice of the one batch
    if os.environ.get("SHN_BENCH_LAUNCH_PROBE"):
        # CPU test of the launcher (tests/test_bench_launcher.py): the ranks meet over gloo, rank 0 prints what the launch resolved
        # to; nothing touches the GPU.  "fail": rank 1 exits non-zero -- the launcher must report it.
        import torch.distributed as dist
        if os.environ["SHN_BENCH_LAUNCH_PROBE"] == "fail" and rank == world - 1:
            sys.exit(3)
        if world > 1:
            dist.init_process_group("gloo")
            one = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(one)
            seen = int(one.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            os.write(json_fd, (json.dumps({"metric": "launch probe", "n_gpus": world, "rccl_ranks": seen, "scaling": args.scaling,
                                           "config": args.config, "K": args.K, "genes": args.genes, "reads_of_the_job": total_reads,
                                           "reads_of_rank0": args.reads, "first_chunk": first_chunk, "steps": args.steps,
                                           "warmup": args.warmup}) + "\n").encode())
        return
    dist = None
    # SHN_BENCH_BACKEND=gloo: development aid -- several ranks on ONE GPU (collectives staged through host memory,
    # exchange.coll_device), to exercise the N-rank code path on a 1-GPU box.  Its numbers mean nothing.
    backend = os.environ.get("SHN_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        if backend == "gloo":
            local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    from shannon_amd import device, exchange
    k1 = args.K + 1
    dev = torch.device("cuda", local if world > 1 else 0)
    seed = 20240501
    # Weak scaling keeps the per-gene depth of configs[1]: the N-rank job is N gene families x 10M reads, and every
    # rank holds a 10M-read slice of that mixture (reads are sharded by index, not by gene).  Adding ranks to ONE
    # family instead would multiply its coverage (616,000x at N=8) -- a different, degenerate assembly problem.
    families = args.families if args.families else (world if (args.genes == 1 and world > 1 and args.scaling == "weak") else 0)
    lock = None
    if world > 1 and backend == "gloo" and os.environ.get("SHN_BENCH_SERIALIZE", "1") == "1":
        import fcntl

        class _FileLock(object):          # ranks sharing the GPU compute one at a time: per-stage times as if each had its own
            def __init__(self, path):
                self.f = open(path, "w")

            def acquire(self):
                fcntl.flock(self.f, fcntl.LOCK_EX)

            def release(self):
                fcntl.flock(self.f, fcntl.LOCK_UN)
        lock = _FileLock("/tmp/shn_bench_lock_%s" % os.environ.get("MASTER_PORT", "0"))

    if lock:
        lock.acquire()
    # strong: this rank's chunks of the one batch (read seed shared); weak: a batch of its own per rank
    r1, r2 = gen_reads(args.reads // 2, seed, args.genes, dev, read_seed=seed + 2 + (0 if args.scaling == "strong" else 1000 * rank),
                       families=families, first_chunk=first_chunk, exon_len=exon_len, chain_exons=chain_exons)
    ctx = device.Context(local if world > 1 else 0)
    sets = [device.Reads.from_codes(ctx, r1), device.Reads.from_codes(ctx, r2)]
    n_reads = len(sets[0]) + len(sets[1])
    if lock:
        ctx.sync()
        lock.release()

    from shannon_amd import pipeline, kmers_for_component as kfc
    store = kfc.ReadStore(r1, r2)
    stage_t = {}

    class _Done(object):
        def close(self):
            pass

    use_dist = world > 1 or args.force_distributed
    if use_dist and dist is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from shannon_amd import distributed

    # Every step is checked, not only the last: the transcripts of each timed step leave a checksum of their buffers (names,
    # sequences, offsets in file order -- xxh3 over ~60 MB: a few ms, inside the timed region) and after the timed region every
    # step must equal the first; the first (= every) step is then held to the pinned order-free sha256 of its config
    # (tests/golden/fullsize_digests.json).  SHN_EXT_DIGEST=1 (a diagnostic run: the checksums cost ~0.1 s per step) also keeps the
    # extension's stage checksums of every step, so that a differing step names the first stage that differed.
    step_digests, step_ext = [], []

    def _quick(final):
        if final is None:
            return None
        if hasattr(final, "quick_digest"):
            return final.quick_digest()
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        for nm, sq in final.items():
            h.update(nm.encode() + b"\t" + sq.encode() + b"\n")
        return h.hexdigest()

    def step():
        if use_dist:
            ops = distributed.GpuOps(ctx, sets[0], sets[1], store, args.K)
            res = distributed.assemble_distributed(ops, args.K, 500, "bench", 1, timings=stage_t, lock=lock)
            d = _Done()
            d.res = res
            step_digests.append(_quick(res["final"]) if res is not None else None)
            return d
        R = pipeline.assemble_resident(ctx, sets[0], sets[1], store, K=args.K, sample="bench", seed=1, timings=stage_t, keep_partitioning=True)
        d = _Done()
        d.R = R
        step_digests.append(_quick(R.final))
        step_ext.append(getattr(R.extension, "ext_digests", None))
        return d

    for _ in range(args.warmup):
        step().close()
    warm_digests = list(step_digests)
    del step_digests[:], step_ext[:]
    stage_t.clear()                      # host stage times: timed steps only
    ctx.timer_reset()
    exchange.stats_reset()
    ctx.lp_stats(reset=True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.time()
    last = None
    for _ in range(args.steps):
        if last is not None:
            last.close()
        last = step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.time() - t0
    # ---- every timed step against the first (and the warm-up steps against it too): a run that differs from itself has no number
    if rank == 0:
        ref_dg = step_digests[0] if step_digests else None
        bad = [i for i, dg in enumerate(step_digests) if dg != ref_dg]
        bad_warm = [i for i, dg in enumerate(warm_digests) if dg != ref_dg]
        if bad or bad_warm:
            sys.stderr.write("bench.py: the transcripts of timed step(s) %s / warm-up step(s) %s differ from timed step 0\n  timed: %s\n  warm-up: %s\n"
                             % (bad, bad_warm, step_digests, warm_digests))
            for i in bad:
                a, b = (step_ext[0] if step_ext else None), (step_ext[i] if i < len(step_ext) else None)
                if a is not None and b is not None:
                    import numpy as _np
                    names = ["table keys", "table counts", "bucket offsets", "weights + flags", "adjacency records", "seed order", "converged claims", "walk records"]
                    st = [j for j in range(8) if not _np.array_equal(a[j], b[j])]
                    sys.stderr.write("  step %d: shn_ext_digests differ first at stage %s (chunks %s of 64); stages that differ: %s\n"
                                     % (i, names[st[0]] if st else "none -- the extension agreed, a later stage differed",
                                        _np.nonzero(a[st[0]] != b[st[0]])[0].tolist() if st else [], [names[j] for j in st]))
                    sys.stderr.write("  step %d shn_ext_digests: %s\n" % (i, b.tolist()))
                else:
                    sys.stderr.write("  step %d: no stage checksums (run with SHN_EXT_DIGEST=1 to have shn_ext_digests of every step)\n" % i)
            sys.stderr.flush()
            os._exit(3)
    tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if (dist and dist.get_backend() == "gloo") else dev)
    if dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    job_reads = n_reads
    if dist and world > 1:                 # reads of the whole job (the ranks' slices of a strong-scaling batch differ by a chunk)
        nt = torch.tensor([n_reads], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(nt, op=dist.ReduceOp.SUM)
        job_reads = int(nt.item())
    timers = ctx.timers()
    lp = ctx.lp_stats()
    if dist and world > 1:                 # the owners of the partitions run the sparse flow: census over all ranks
        keys = [k for k in sorted(lp) if k != "rule"]
        lt = torch.tensor([lp[k] for k in keys], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(lt, op=dist.ReduceOp.SUM)
        lp.update({k: int(v) for k, v in zip(keys, lt.tolist())})
    stage_max = None
    coll = None
    if use_dist:
        coll = exchange.stats_snapshot()
    if dist and world > 1:                 # slowest rank per stage (the stage dict of rank 0 alone hides imbalance)
        allst = [None] * world
        dist.all_gather_object(allst, dict(stage_t))
        stage_max = {k: max(d.get(k, 0.0) for d in allst) / args.steps for k in allst[0]}
    if use_dist:
        distinct, total = (last.res["n_k1mers"] if rank == 0 else 0), None
    else:
        distinct, total = last.R.n_k1mers, last.R.n_windows
    if rank == 0:
        ms_step = 1000.0 * dt / args.steps
        # Every timer below wraps exactly ONE kernel launch (HIP events on the stream it is launched on, shn_timer_*),
        # so ms / launches is that kernel's average launch duration -- comparable with rocprofv3's AverageNs.
        # Algorithmic bytes per unit are the figures of DESIGN.md section 3.
        W = 100 - k1 + 1
        KERNEL = {"count.direct": "count_direct_kernel", "count.hist1": "hist1_kernel", "count.scatter1": "scatter1_kernel", "count.hist2": "hist_keys_kernel",
                  "count.scatter2": "scatter_keys_kernel", "count.buckets": "buckets_kernel", "route": "route_kernel",
                  "extend.walk_fresh": "ext_walk_kernel<true>", "extend.walk_thread": "ext_walk_kernel<false>", "extend.walk_wave": "ext_walk_long_kernel",
                  "extend.mark": "ext_mark_kernel", "extend.begin": "ext_round_begin_kernel",
                  "extend.adjacency": "ext_records_kernel",
                  # the super-k-mer counting path (csrc/count_sk.hip)
                  "count.sk_emit": "sk_scan_kernel", "count.sk_hist": "skr_hist_kernel + skr_scatter_kernel (level 1)", "count.sk_hist2": "skr_hist_kernel",
                  "count.sk_scatter2": "skr_scatter_kernel", "count.sk_buckets": "sk_buckets_sorted_kernel<.., 256, 1024, 0>",
                  "count.sk_buckets2": "sk_buckets_sorted_kernel<.., 256, 2048, 0>",
                  # round 6: the contig stage, the read -> graph mapping and the LP trials -- their algorithmic bytes are declared by the
                  # launch sites (TimerRegion::bytes, shn_timer_bytes: the byte model stands next to the launch in csrc/)
                  "contig.sort": "cg_keys_kernel + sort_hist_kernel + sort_scatter_kernel (shn_sorted_windows)", "contig.hits": "cg_hits_kernel",
                  "contig.cover": "cg_cover_kernel", "contig.compact": "cg_acc_compact_kernel", "graph.kp_search": "kp_search_all",
                  "graph.kp_classify": "kp_classify", "graph.seed_scan": "seed_scan_reads_kernel", "graph.dd_insert": "dd_insert",
                  "lp.trials": "lp_trials_coop_kernel", "extend.audit": "ext_audit_nodes_kernel + ext_audit_walks_kernel"}
        # records of the super-k-mer path: a read of W windows makes ~2 W / (w + 1) + 1 records of 16 bytes (w = k1 - m + 1 m-mers per window, m = 13)
        sk_w = k1 - max(13, 2 * k1 - 48) + 1
        rec_bytes = 16.0 * (2.0 * W / (sk_w + 1) + 1.0)
        per_read = {"count.direct": 25.0 + 12.0 * distinct / max(1, n_reads),    # packed read in, (key, count) of the distinct k1-mers out
                    "count.hist1": 25.0, "count.scatter1": 25.0 + 8.0 * W, "count.hist2": 8.0 * W,
                    "count.scatter2": 16.0 * W, "count.buckets": 8.0 * W, "route": 192.0,
                    "count.sk_emit": 25.0 + rec_bytes, "count.sk_hist": 2.0 * rec_bytes, "count.sk_hist2": rec_bytes, "count.sk_scatter2": 2.0 * rec_bytes,
                    "count.sk_buckets": rec_bytes + 12.0 * distinct / max(1, n_reads)}
        ext = last.res["extension"] if use_dist else {k: getattr(last.R.extension, k, None) for k in ("iterations", "n_walks", "total_steps", "wave_steps", "fresh_steps", "dense_rounds", "settled_walks")}
        steps_all = ext["total_steps"] or 0
        steps_wave = ext["wave_steps"] or 0
        steps_fresh = ext.get("fresh_steps") or 0
        n_or = 2 * distinct                                   # oriented k1-mers
        # bytes of ONE step of the pipeline (all launches of the kernel in that step)
        per_step_bytes = {k: v * n_reads for k, v in per_read.items()}
        # a thread-walker step: 4 candidates x (claim 8 + snapshot 8 + weight 4 + row 16) + claim 8 = 152 B; the first round of a rank block
        # (ext_walk_kernel<true>, its own timer) reads no snapshot: 120 B
        per_step_bytes["extend.walk_fresh"] = 120.0 * steps_fresh
        per_step_bytes["extend.walk_thread"] = 152.0 * (steps_all - steps_wave - steps_fresh)
        per_step_bytes["extend.begin"] = 8.0 * n_or * (timers.get("extend.begin", (0, 0))[1] / max(1, args.steps))   # the claims, once per launch (the three launches that also copy the snapshot move 24 B per k1-mer)
        per_step_bytes["count.sk_buckets2"] = 0.07 * per_step_bytes["count.sk_buckets"]                                 # (7 % of the buckets do not fit the small table and are read again)
        per_step_bytes["extend.walk_wave"] = 112.0 * steps_wave                      # same without the row prefetch, + memo entry
        # mark pass: claim + snapshot of every oriented k1-mer in the rounds that stream them (dense), one flag byte per 16 k1-mers in the others
        dense_r = ext.get("dense_rounds") if ext.get("dense_rounds") is not None else (ext["iterations"] or 0)
        per_step_bytes["extend.mark"] = 16.0 * n_or * dense_r + (n_or / 16.0) * max(0, (ext["iterations"] or 0) - dense_r)
        per_step_bytes["extend.adjacency"] = (8 * 8.0 + 2 * 64.0 + 8.0) * distinct       # per k1-mer: 8 keys looked up, its two 64-byte records written, its own key
        # the always-on fixpoint audit: the claims streamed once (8 B per oriented k1-mer) + per claimed k1-mer its step's lines; priced on the stream alone
        per_step_bytes["extend.audit"] = 8.0 * n_or
        # the slots whose launch sites declare their bytes themselves
        tb = ctx.timer_bytes()
        for k_, b_ in tb.items():
            if k_ in KERNEL and k_ not in per_step_bytes:
                per_step_bytes[k_] = b_ / float(args.steps)
        kt = {k: v for k, v in timers.items() if k in KERNEL and k in per_step_bytes}
        # the key array goes through one or two levels below level 1 (csrc/count.hip: at most 256 streams per level): the bytes of
        # the hist2 / scatter2 launches of a step are those of as many passes over the keys
        if "count.buckets" in timers and "count.scatter2" in timers:
            levels = max(1.0, timers["count.scatter2"][1] / max(1, timers["count.buckets"][1]))
            per_step_bytes["count.hist2"] *= levels
            per_step_bytes["count.scatter2"] *= levels

        def roof(name):
            ms, launches = kt[name]
            avg = ms / launches
            bytes_launch = per_step_bytes[name] * args.steps / launches
            ach = bytes_launch / (avg * 1e-3) / 1e9
            tr = TRAFFIC.get("config%s" % args.config, {}).get(name) if (is_config and world == 1) else None   # PMC passes were taken at N=1
            return {"bound": "hbm", "kernel": KERNEL[name], "timer": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": tr, "launches_per_step": launches / args.steps, "avg_launch_ms": avg,
                    "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_read": per_read.get(name)}
        dom = max(kt, key=lambda k: kt[k][0])
        bdom = max((k for k in kt if k in per_read), key=lambda k: kt[k][0])
        r_dom, r_bw = roof(dom), roof(bdom)
        # every timed kernel against the HBM roof (algorithmic bytes of DESIGN.md section 3 / SURVEY 8d, time from the HIP-event timers)
        kernel_table = []
        for name in sorted(kt, key=lambda k: -kt[k][0]):
            q = roof(name)
            kernel_table.append({"kernel": q["kernel"], "timer": name, "ms_per_step": kt[name][0] / args.steps, "launches_per_step": q["launches_per_step"],
                                 "algorithmic_GB_per_step": per_step_bytes[name] / 1e9, "algorithmic_bytes_per_read": per_step_bytes[name] / max(1, n_reads),
                                 "achieved_GBs": q["achieved"], "frac_of_hbm_peak": q["frac"], "traffic_bytes_per_launch_pmc": q["traffic"]})
            if q["traffic"]:
                # how busy the memory system is with this kernel, its waste included: PMC traffic per launch / the launch's time, against
                # the 8 TB/s of streams and -- as 128-byte requests -- against the 48.4 G/s of independent random loads (a gather's roof)
                tg = q["traffic"] / (q["avg_launch_ms"] * 1e-3) / 1e9
                kernel_table[-1].update({"traffic_GBs_pmc": tg, "traffic_frac_of_hbm_peak": tg / HBM_PEAK_GBS,
                                         "requests_frac_of_random_access_roof": tg / 128.0 / 48.4})
        # the whole path against the HBM roof on the survey's byte model -- 2.5 KB per 100 bp read (1.23 KB of it counting)
        BYTES_PER_READ_8D = 2500.0
        e2e_gbs = job_reads * args.steps / dt * BYTES_PER_READ_8D / 1e9
        roofline_e2e = {"bound": "hbm", "achieved": e2e_gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": e2e_gbs / (HBM_PEAK_GBS * world),
                        "algorithmic_bytes_per_read": BYTES_PER_READ_8D,
                        "note": "reads/s x 2.5 KB per read (SURVEY 8d byte model) / (n_gpus x 8 TB/s); the path is not HBM-bound end to end at this "
                                "byte model -- the per-kernel fractions are in kernel_table"}
        if dom.startswith("extend."):
            r_dom["note"] = ("greedy walk fixpoint: dependent pointer chasing, one memory round trip per step -- latency-bound, not "
                             "bandwidth-bound; the dominant streaming kernel is in roofline_bandwidth_kernel")
            # the same kernel against the roof that binds a gather: every random load is one 128-byte request to HBM whatever it uses,
            # and the chip serves 48.4 G of them per second when nothing depends on anything (profiles/r03_fetch_calibration.txt: 1.074 G
            # random 8-byte loads in 22.2 ms).  Requests per launch = PMC traffic / 128.
            if r_dom.get("traffic"):
                req = r_dom["traffic"] / 128.0
                r_dom["random_access"] = {"requests_per_launch": req, "achieved_G_requests_per_s": req / (r_dom["avg_launch_ms"] * 1e-3) / 1e9,
                                          "peak_G_requests_per_s": 48.4, "frac": req / (r_dom["avg_launch_ms"] * 1e-3) / 1e9 / 48.4,
                                          "note": "independent random loads; a walk's steps depend on one another"}
        what = {"1": "10M synthetic 2x100bp paired reads, k=25 (k1=26), single gene family, 0.5% substitution errors (BASELINE configs[1])",
                "2": "100M synthetic 2x100bp paired reads (50M pairs), k=25 (k1=26), 20,000 genes (1-6 isoforms of 3-12 exons, lognormal "
                     "expression), 0.5% substitution errors, multi-component, --partition 500 (BASELINE configs[2])",
                "4s": "100M synthetic 2x100bp paired reads (50M pairs), k=31 (k1=32), 4,000 genes with exons of 80-5,000 bp (long unitigs), "
                      "0.5% substitution errors, --partition 500 (one-GPU slice of BASELINE configs[4]: a fifth of its reads and genes)",
                "2p": "20M synthetic 2x100bp paired reads (10M pairs), k=25 (k1=26), 30,000 genes linked by 6,000 shared repeat elements in a chain "
                      "(one contig-graph component per strand with about two contigs per gene, cut into 100 parts twice by the library's "
                      "partitioner -- row a8, kmers_for_component.py:207-237), --partition 500; none of BASELINE's configs"}[args.config]
        if not is_config and not families:
            workload = ("%d synthetic 2x100bp paired reads%s, K=%d, %d genes (a tuning input, none of BASELINE's configs)"
                        % (total_reads, " per GPU" if (world > 1 and args.scaling == "weak") else "", args.K, args.genes))
        elif world > 1 and args.scaling == "strong":
            workload = ("%s hash-sharded across %d GPUs with the RCCL all-to-all bucket exchange: every rank holds a contiguous slice of the "
                        "batch's chunks%s" % (what, world, " (BASELINE configs[3])" if args.config == "2" else ""))
        elif families > 1:
            workload = ("10M synthetic 2x100bp paired reads per GPU, k=25 (k1=26), %d gene families, one per rank's worth of reads, every rank "
                        "holding a slice of the mixture, 0.5%% substitution errors (BASELINE configs[1] per family)" % families)
        elif world > 1:
            workload = what + ", one such batch per GPU (weak scaling)"
        else:
            workload = what
        out = {
            "metric": "reads/sec k-mer->graph->path-decompose, 2x100bp k=%d" % args.K,
            "value": job_reads * args.steps / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload,
                       "baseline_config": (("3" if (args.config == "2" and world > 1 and args.scaling == "strong") else args.config) if (is_config and args.config != "2p") else None),
                       "reads_per_gpu": n_reads, "reads_of_the_job": job_reads, "K": args.K,
                       "rccl_ranks": (dist.get_world_size() if dist else 1), "collective_backend": (dist.get_backend() if dist else None),
                       "collectives_per_step": ({k: {"calls": v["calls"] / args.steps, "bytes_sent": v["bytes_sent"] / args.steps,
                                                     "bytes_received": v["bytes_received"] / args.steps, "seconds": v["seconds"] / args.steps}
                                                 for k, v in coll.items()} if coll is not None else None),
                       "stages": ("full path a1-a31, sharded: local count -> all-to-all bucket exchange -> replicated extension -> local routing -> "
                                  "owner-side graph + sparse flow -> gather + merge on rank 0" if use_dist else
                                  "full path a1-a31: count -> extension -> partition/route -> multibridged graph -> sparse flow -> merge"),
                       "host_stage_seconds_per_step": {k: v / args.steps for k, v in stage_t.items()},
                       "host_stage_seconds_per_step_slowest_rank": stage_max,
                       "transcripts": (len(last.res["final"]) if use_dist else len(last.R.final)),
                       "transcripts_sha256_16": _final_sha(last.res["final"] if use_dist else last.R.final),
                       "steps_checked": {"timed": len(step_digests), "warmup": len(warm_digests), "all_equal": True, "checksum": step_digests[0] if step_digests else None,
                                         "pinned_sha256_16": None},
                       # row a28 census: path_decompose calls that reached the LP trials, and those of them in which the optimal face of a
                       # trial was not a point (there the interior-point limit -- the analytic centre -- differs from a vertex)
                       "lp_rule": lp["rule"], "lp_calls": lp["lp_calls"] / args.steps, "lp_degenerate": lp["lp_degenerate"] / args.steps,
                       "lp_trials": lp["lp_trials"] / args.steps, "lp_degenerate_trials": lp["lp_degenerate_trials"] / args.steps,
                       "lp_newton_steps": lp["newton_steps"] / args.steps, "lp_not_converged": lp["not_converged"], "lp_too_large_trials": lp["too_large_trials"],
                       "extension_iterations": ext["iterations"], "extension_walks": ext["n_walks"],
                       "extension_walk_steps": steps_all, "extension_dense_rounds": ext.get("dense_rounds"),
                       "extension_walks_settled_by_chain": ext.get("settled_walks"),
                       "partitions": (len(last.res["partitions"]) if use_dist else {k: [v["n_reads_routed"], v["n_k1mers"]] for k, v in last.R.partitions.items()}),
                       "windows_per_step": total, "distinct_k1mers": distinct},
            "roofline": r_dom,
            "roofline_bandwidth_kernel": r_bw,
            "roofline_e2e": roofline_e2e,
            # the counting stage as a whole on SURVEY 8d's counting bytes (read the packed read once, emit every k1-mer key once and read
            # it once: 25 + 16 W bytes per read), time = the stage's event timer
            "roofline_count_stage": ({"bound": "hbm", "timer": "count.total", "algorithmic_bytes_per_read": 25.0 + 16.0 * W,
                                      "achieved": (25.0 + 16.0 * W) * n_reads / (timers["count.total"][0] / args.steps * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": (25.0 + 16.0 * W) * n_reads / (timers["count.total"][0] / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "ms_per_step": timers["count.total"][0] / args.steps} if "count.total" in timers else None),
            "kernel_table": kernel_table,
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in sorted(timers.items())},
            "kernel_launches_per_step": {k: v[1] / args.steps for k, v in sorted(timers.items())},
        }
        # the pinned digest of the configuration (the one the full-size GPU tests check): a run whose every step agrees with itself
        # but not with the pin fails too
        try:
            pins = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))
        except OSError:
            pins = {}
        pin = pins.get(args.config) if is_config else None
        out["config"]["steps_checked"]["pinned_sha256_16"] = pin
        if pin is not None and pin != out["config"]["transcripts_sha256_16"] and os.environ.get("SHN_BENCH_ALLOW_UNPINNED") != "1":
            sys.stderr.write("bench.py: transcripts digest %s of --config %s differs from the pinned %s (tests/golden/fullsize_digests.json)\n"
                             % (out["config"]["transcripts_sha256_16"], args.config, pin))
            sys.stderr.flush()
            os._exit(4)
        if world == 1 and not use_dist:
            # ingest-inclusive rate: the resident step plus reading the batch from FASTA text at the measured ingest rate (no overlap
            # between ingest and the step assumed)
            ing = ingest_rate(ctx, r1, r2, min(len(r1), 5_000_000))
            out["ingest"] = ing
            out["value_with_ingest"] = 1.0 / (1.0 / out["value"] + 1.0 / ing["reads_per_s"])
        n_ov = args.overlap_steps if args.overlap_steps >= 0 else (6 if (is_config and args.config == "2") else 0)
        if world == 1 and not use_dist and n_ov > 0:
            out["overlap"] = overlapped_steps(ctx, sets, store, args.K, n_ov, out["config"]["transcripts_sha256_16"], n_reads)
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 of the N=1 run only
            # bounded sample: ~10-15 s of one core through the whole path (at configs[2] a read costs the pure-Python path ~5x more
            # than at configs[1]: nearly every k1-mer of a 25k-read sample is new), + ~5 s for the counting stage alone / threaded
            # ~30 s of host work: 1 M reads (500,000 pairs) through the whole path, native where a native host form exists.
            # (Round 4 stopped at 0.5 M reads: 28 of its 35 s were the dictionary of oracle/ext_c.c -- a qsort of 42 M entries and a
            # 25-step bisection per look-up; with a radix sort and a prefix table the extension is a few seconds at a million reads.)
            out["cpu_baseline"] = cpu_whole_path_native(args.K, r1, r2, min(len(r1), 500_000))
            # the pure-Python port (oracle/pipeline.py: pure Python like Shannon itself) on a small sample, for scale
            out["cpu_baseline"]["python_port"] = cpu_baseline(k1, r1, r2, 25_000 if args.config == "1" else 6_000)
        final_line = json.dumps(out)
    else:
        final_line = None
    last.close()
    ctx.close()
    if dist:
        dist.destroy_process_group()
    if final_line is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.write(json_fd, (final_line + "\n").encode())


if __name__ == "__main__":
    main()
