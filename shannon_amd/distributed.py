"""Multi-GPU hot path on one node: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).  DESIGN.md section 6.

No rank holds the job: not its reads (every rank ingests its share of the files' BYTES: ingest_rank_slice), not its k1-mer table.
  1. every rank counts its slice of the reads; the (key, count) pairs go to the rank that owns the k1-mer's MINIMIZER in ONE
     all-to-all(v) -- the k-mer bucket exchange -- and the owner reduces by key (a k1-mer and its neighbours share their
     minimizer 7 times out of 8: most edges of the k1-mer graph stay inside a shard);
  2. the connected components of the k1-mer graph are labelled ON the owner shards (local union-find; the keys a higher rank owns
     are asked there in one all-to-all; the edges between local components are all-gathered and solved on every rank alike), and
     whole components travel to the rank that walks them (one all-to-all): a table of its own per rank, the unsharded greedy
     extension on it (SHN_OWNER_LABELS=0: the replicated table of rounds 2-4, kept for comparison);
  3. the candidates of all ranks meet in the walk order of the reference (one all-gather of arrays: seed weights and keys, offsets,
     text), the contig stage runs on them, every rank routes ITS reads against the accepted contigs' probe table; partitions are
     dealt to ranks by the reads their graphs consume (deal_partitions) and those reads -- the first 10 * nodes + 1 of the global
     strand-doubled order (multibridging.py:26-30) -- travel to the owner in one all-to-all of bytes;
  4. owners build the multibridged graphs and run the sparse flow of their partitions; rank 0 gathers the FASTA and merges.
The result equals the single-GPU pipeline on the concatenated reads (tests/test_distributed_*.py, test_nrank_midsize_gpu.py).

`ops` abstracts the per-rank compute (GpuOps: HIP kernels through the C ABI; the CPU tests plug an oracle-backed implementation
to exercise the collective choreography with gloo).
"""
from collections.abc import Mapping
import os
import numpy as np
import torch
import torch.distributed as dist
from . import exchange, mbgraph, sparse_flow, post, _lib


def _all_gather_var(t, group=None, name="table all-gather (owned k1-mer shards)"):
    """all-gather of 1-D tensors of different lengths (padded to the max)."""
    import time
    t_begin = time.time()
    out, ns = _all_gather_var_(t, group)
    if dist.get_world_size(group) > 1:
        if out.device.type == "cuda":
            torch.cuda.synchronize()
        me = dist.get_rank(group)
        es = t.element_size()
        exchange.note(name, es * ns[me] * (len(ns) - 1), es * (sum(ns) - ns[me]), time.time() - t_begin)
    return out, ns


def _all_gather_var_(t, group=None):
    W = dist.get_world_size(group)
    cdev = exchange.coll_device(t.device, group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=cdev)
    ns = [torch.zeros_like(n) for _ in range(W)]
    dist.all_gather(ns, n, group=group)
    ns = [int(x.item()) for x in ns]
    if W == 1:
        return t, ns
    # in rounds of at most exchange.chunk_elems() elements per rank (see there), each round padded to its longest piece
    C = exchange.chunk_elems()
    out = torch.empty(sum(ns), dtype=t.dtype, device=cdev)
    off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    src = t.to(cdev)
    for j in range((max(ns + [1]) + C - 1) // C):
        lens = [int(min(max(k - j * C, 0), C)) for k in ns]
        mx = max(lens + [1])
        pad = torch.zeros(mx, dtype=t.dtype, device=cdev)
        mine = lens[dist.get_rank(group)]
        pad[:mine] = src[j * C: j * C + mine]
        outs = [torch.empty_like(pad) for _ in range(W)]
        dist.all_gather(outs, pad, group=group)
        for r in range(W):
            if lens[r]:
                out[int(off[r]) + j * C: int(off[r]) + j * C + lens[r]] = outs[r][:lens[r]]
    return out.to(t.device), ns


def ingest_rank_slice(paths, rank, world, group, device, ctx=None, stats=None):
    """The N-rank CLI's reads by BYTES of the files (round 6; the reference streams its read files once, 10 M reads at a time:
    kmers_for_component.py:322-403).  Every rank counts the records that start in its share of each file's bytes
    (shn_text_records_in_range), the counts of all ranks say which record every share starts with, and the records
    [n r / W, n (r + 1) / W) of this rank -- a few records into some share -- go through shn_reads_ingest as one stretch of text.
    Host memory and parsing per rank: its share of the text (memory-mapped) + its slice's code matrix, not the whole job's.
    Returns (code matrices of this rank's slice, one per file; n records of the job) or None when a file cannot be shared out this
    way (.gz: no random access; reads of different lengths; multi-line FASTA) and the caller reads whole files as before.
    stats (dict): bytes this rank looked at / holds."""
    import ctypes as C
    from . import _lib, device as dev_mod
    if any(p.endswith(".gz") for p in paths):
        return None
    L_ = _lib.lib()
    mats, n_job = [], None
    scanned = held = 0
    for path in paths:
        size = os.path.getsize(path)
        text = np.memmap(path, dtype=np.uint8, mode="r") if size else np.zeros(0, np.uint8)
        B = int(len(text))
        lo_b, hi_b = rank * B // world, (rank + 1) * B // world
        first, cnt, n_idx = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ptr = text.ctypes.data if B else None
        ok = 1
        # (with the count: the offset of every 1,024th record of the share -- from the shares' sparse indices any record of the file is
        # at most 1,023 records away, whoever's share it starts in)
        STRIDE = 1024
        idx_buf = np.zeros(max(1, (hi_b - lo_b) // (2 * STRIDE) + 2), dtype=np.uint64)         # (a record is at least 2 bytes)
        try:
            _lib.check(L_.shn_text_records_in_range(ptr, B, lo_b, hi_b, 0, C.byref(first), C.byref(cnt), STRIDE, idx_buf.ctypes.data, len(idx_buf),
                                                    C.byref(n_idx)))
        except _lib.ShannonError:
            ok = 0
        scanned += hi_b - lo_b
        sparse = [p_[0] for p_ in exchange.all_gather_arrays((idx_buf[:n_idx.value],), device, group, "ingest: sparse record index of the shares")]
        mine = torch.tensor([first.value, cnt.value, ok], dtype=torch.int64, device=device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        tab = np.stack([x.cpu().numpy() for x in parts])                      # [W, 3]: first record start, records, ok
        if not tab[:, 2].all():
            return None
        firsts, cnts = tab[:, 0], tab[:, 1]
        prefix = np.concatenate([[0], np.cumsum(cnts)])
        n = int(prefix[-1])
        if n_job is None:
            n_job = n
        elif n != n_job:
            raise _lib.ShannonError("--left and --right hold different numbers of reads (%d, %d)" % (n_job, n))

        def locate(idx):
            # byte offset at which record `idx` of the file starts
            nonlocal scanned
            if idx >= n:
                return B
            k = int(np.searchsorted(prefix, idx, side="right")) - 1
            j, rest = divmod(int(idx - prefix[k]), STRIDE)
            start = int(sparse[k][j])
            off = C.c_uint64()
            _lib.check(L_.shn_text_skip_records(ptr, B, start, rest, 0, C.byref(off)))
            scanned += int(off.value) - start
            return int(off.value)
        rlo, rhi = rank * n // world, (rank + 1) * n // world
        a, b = locate(rlo), locate(rhi)
        if rhi == rlo:                                        # (more ranks than records: this rank holds none)
            codes = np.zeros((0, 0), np.uint8)
        else:
            try:
                _d, codes = dev_mod.Reads.ingest(None, text[a:b])
            except _lib.ShannonError as ex:
                if "unsupported" not in str(ex):
                    raise
                codes = None
        fine = codes is not None and not isinstance(codes, dev_mod.RaggedCodes) and len(codes) == rhi - rlo
        flag = torch.tensor([1 if fine else 0, codes.shape[1] if fine and len(codes) else 0], dtype=torch.int64, device=device)
        flags = [torch.zeros_like(flag) for _ in range(world)]
        dist.all_gather(flags, flag, group=group)
        ft = np.stack([x.cpu().numpy() for x in flags])
        lens = set(int(v) for v in ft[:, 1] if v)
        if not ft[:, 0].all() or len(lens) > 1:
            return None
        if len(codes) == 0 and lens:                                          # (a rank without records still holds a matrix of the job's read length)
            codes = np.zeros((0, lens.pop()), np.uint8)
        scanned += b - a
        held += (b - a) + codes.size
        mats.append(np.ascontiguousarray(codes))
        del text
    if stats is not None:
        stats.update({"bytes_scanned": int(scanned), "bytes_held": int(held), "file_bytes": int(sum(os.path.getsize(p) for p in paths)), "records": int(n_job or 0)})
    return mats, int(n_job or 0)


def deal_partitions(load, W):
    """owner[i] of partition i: partitions in order of decreasing load, each to the rank with the least load so far (ties: the
    lowest rank, the lowest partition index first) -- identical on every rank."""
    load = np.asarray(load, dtype=np.int64)
    owner = np.zeros(len(load), dtype=np.int64)
    acc = np.zeros(W, dtype=np.int64)
    for i in np.argsort(-load, kind="stable").tolist():
        r = int(np.argmin(acc))
        owner[i] = r
        acc[r] += load[i]
    return owner


class _NoLock(object):
    def acquire(self):
        pass

    def release(self):
        pass


def assemble_distributed(ops, K=25, partition_size=500, sample="shannon", seed=0, part_vectors=None, group=None,
                         timings=None, lock=None, double_stranded=True, min_weight=3, min_length=75, kmer_hard_cutoff=1):
    """Returns on rank 0 a dict {partitions: {name: fasta}, final, contigs, ...}; None elsewhere.
    min_weight / min_length: hyp_min_weight (--kmer_soft_cutoff) / hyp_min_length as run_correction gets them (shannon.py:243-247,
    457); kmer_hard_cutoff: `jellyfish dump -L` (shannon.py:237-241, 441), applied by the ops to the REDUCED counts -- on the owner
    of a k1-mer after the bucket exchange, never to a rank's local counts.  The ops read them as attributes of the same names.
    double_stranded=False (-s / --ss / --strand_specific, shannon.py:407-411): the reads are not strand-doubled -- forward counting
    of reads_1 and RC(reads_2), routes of plain read indices (the global order is the global read index, there is no
    reverse-complement half), pairs (reads_1[i], RC(reads_2[i])) at the owners, process_concatenated_fasta with the user's flag
    (:596).  The ops say whether they can (`ops.strand_specific`, set here); ops without the attribute `supports_strand_specific`
    are refused rather than run double-stranded.
    timings: seconds per stage, compute ("count", "extension", ...) and collectives ("x:...") apart.
    lock (development aid): held while this rank computes, released around every collective -- with several ranks on
    ONE GPU it serialises the compute, so the per-stage compute times are those of a rank that has a GPU to itself."""
    import time
    ss = not double_stranded
    if ss and not getattr(ops, "supports_strand_specific", False):
        raise NotImplementedError("assemble_distributed: these ops do not implement -s / --strand_specific (forward counting, plain read "
                                  "indices, pairs of reads_1 and RC(reads_2)); a strand_specific run needs ops that do (GpuOps does)")
    ops.strand_specific = ss
    ops.min_weight, ops.min_length, ops.kmer_hard_cutoff = int(min_weight), int(min_length), int(kmer_hard_cutoff)
    T = timings if timings is not None else {}
    lock = lock or _NoLock()
    ops.lock = lock
    ops.timings = T                     # sub-stages of the extension land in the same dict ("ext.*")

    def tick(name, t0):
        T[name] = T.get(name, 0.0) + time.time() - t0

    W, rank = dist.get_world_size(group), dist.get_rank(group)
    # ---- 1. local count + bucket exchange + reduce by key
    lock.acquire()
    t0 = time.time()
    import os
    forced = os.environ.get("SHN_OWNER_LABELS") == "2" and getattr(ops, "owner_labelling", False)    # (tests: the N-rank path on one rank)
    if W == 1 and hasattr(ops, "local_table") and not forced:
        # one rank: nothing to exchange, reduce or gather -- the counted table IS the table (it used to be exported to pairs, reduced
        # into a table, exported again and rebuilt: two passes through the pairs path, 0.2 s at configs[2])
        table = ops.local_table()
        tick("count", t0)
        gk = None
    elif getattr(ops, "owner_labelling", False):
        # No rank ever holds the whole table (BASELINE configs[4]): the k1-mers go to the rank that owns their minimizer, the components
        # of the k1-mer graph are labelled on those shards (local union-find + one exchange of the edges that cross shards), and whole
        # components travel to the rank that walks them -- a table of its own per rank, the unsharded extension on it.
        keys, counts, send = ops.local_pairs(W, by_minimizer=True)
        tick("count", t0)
        lock.release()
        t0 = time.time()
        rk, rc, _ = exchange.all_to_all_pairs(keys, counts, send, group)
        del keys, counts
        tick("x:bucket exchange", t0)
        lock.acquire()
        t0 = time.time()
        owned = ops.owned_table(rk, rc)
        del rk, rc
        tick("reduce", t0)
        table, n_table = ops.component_table(owned, group, tick)
        gk = None
        t0 = time.time()
    else:
        keys, counts, send = ops.local_pairs(W)
        tick("count", t0)
        lock.release()
        t0 = time.time()
        rk, rc, _ = exchange.all_to_all_pairs(keys, counts, send, group)
        tick("x:bucket exchange", t0)
        lock.acquire()
        t0 = time.time()
        ok, oc = ops.reduce_pairs(rk, rc)
        tick("reduce", t0)
        lock.release()
        # ---- 2. replicate the (small) distinct-k1-mer table, extension + partitioning on every rank
        t0 = time.time()
        gk, _ = _all_gather_var(ok, group)
        gc, _ = _all_gather_var(oc, group)
        tick("x:allgather table", t0)
        lock.acquire()
        t0 = time.time()
        table = ops.table_from_pairs(gk, gc)
        tick("table", t0)
    t0 = time.time()
    if (W > 1 or forced) and getattr(ops, "owner_labelling", False):
        res = ops.extension(table, partition_size, group, presharded=n_table)
    else:
        n_table = len(table) if gk is None else None
        res = ops.extension(table, partition_size, group) if getattr(ops, "sharded_extension", False) else ops.extension(table, partition_size)
    if n_table is not None:
        res.n_k1mers_table = n_table
    tick("extension", t0)
    w = getattr(ops, "coll_wait", 0.0)       # the sharded contig stage's object collectives, reported apart
    if w:
        T["extension"] -= w
        T["x:contig gathers"] = T.get("x:contig gathers", 0.0) + w
    t0 = time.time()
    part = ops.route(res, K, partition_size, part_vectors)
    names = list(part["new_components"])
    P = len(names)
    # ---- 3. global strand-doubled order + per-partition cap
    n_local = ops.n_reads()
    tick("partition+route", t0)
    lock.release()
    t0 = time.time()
    cdev = exchange.coll_device(ops.device, group)
    nl = torch.tensor([n_local], dtype=torch.int64, device=cdev)
    nls = [torch.zeros_like(nl) for _ in range(W)]
    dist.all_gather(nls, nl, group=group)
    nls = [int(x.item()) for x in nls]
    base, n_glob = sum(nls[:rank]), sum(nls)
    cnt = np.zeros((P, 2), dtype=np.int64)          # routed reads of this rank: forward half / RC half
    for i, nm in enumerate(names):
        r = part["routes"][nm]
        f = r.count_below_split() if hasattr(r, "count_below_split") else int(np.searchsorted(r, n_local))
        cnt[i] = (f, len(r) - f)
    ct = torch.as_tensor(cnt.reshape(-1), device=cdev)
    cts = [torch.zeros_like(ct) for _ in range(W)]
    dist.all_gather(cts, ct, group=group)
    allc = np.stack([c.cpu().numpy().reshape(P, 2) for c in cts])       # [W, P, 2]
    tick("x:route counts", t0)
    lock.acquire()
    t0 = time.time()
    # position of this rank's first forward / first RC routed read of partition p in the global order
    fwd_before = allc[:rank, :, 0].sum(axis=0)
    rc_before = allc[:, :, 0].sum(axis=0) + allc[:rank, :, 1].sum(axis=0)
    paired = ops.paired
    # owner of every partition: largest first onto the least loaded rank (load = the reads its graph will consume, known to
    # every rank from the gathered counts) -- the reference's size-sorted job list (shannon.py:546-551) dealt over the ranks
    cutoffs = np.array([10 * ops.n_nodes(part, nm, K) + 1 for nm in names], dtype=np.int64)
    owner = deal_partitions(np.minimum(allc.sum(axis=(0, 2)), cutoffs) + 1, W)
    payload = [[] for _ in range(W)]                # per destination rank: (partition, global doubled indices, reads)
    for i, nm in enumerate(names):
        cutoff = int(cutoffs[i])
        r = part["routes"][nm]                      # array, or a RouteView: only the kept prefixes are fetched
        f = int(cnt[i, 0])
        keep_f = int(max(0, min(f, cutoff - fwd_before[i])))
        keep_r = int(max(0, min(len(r) - f, cutoff - rc_before[i])))
        if keep_f + keep_r == 0:
            continue
        rf, rr = np.asarray(r[:keep_f], dtype=np.int64), np.asarray(r[f:f + keep_r], dtype=np.int64)
        sel = np.concatenate([rf, rr])
        gidx = np.concatenate([base + rf, n_glob + base + (rr - n_local)])
        if not ss and int(owner[i]) == rank and getattr(ops, "resident_rows", False) and int(allc[:, i, :].sum()) == int(cnt[i].sum()):
            # every routed read of this partition is a row of this rank's own resident input, and this rank owns the partition:
            # nothing is collected -- the graph stage names the reads by their rows, as the single-GPU pipeline does
            payload[rank].append((i, gidx, LocalRows(sel)))
        else:
            payload[int(owner[i])].append((i, gidx, ops.collect(sel)))
    tick("collect reads", t0)
    lock.release()
    t0 = time.time()
    if getattr(ops, "array_payload", False):        # code rows + flags: one all-to-all(v) of bytes; what a rank owns itself stays put
        got = exchange.all_to_all_bytes([exchange.pack_read_pieces(items if d != rank else []) for d, items in enumerate(payload)], ops.device, group)
        recv = [exchange.unpack_read_pieces(b) if src != rank else payload[rank] for src, b in enumerate(got)]
    else:                                           # python objects (the oracle-backed test ops): small inputs only
        recv = [None] * W
        _a2a_objects(recv, payload, group)
    tick("x:read exchange", t0)
    # ---- 4. owned partitions: graph + sparse flow
    lock.acquire()
    t0 = time.time()
    mine = {}
    for lst in recv:
        for p, gidx, data in lst:
            mine.setdefault(p, []).append((gidx, data))
    owned = sorted((i for i in range(P) if int(owner[i]) == rank), key=lambda i: -int(min(allc[:, i, :].sum(), cutoffs[i])))

    if hasattr(ops, "graph_batch"):
        # the owned partitions through the single-GPU graph stage: GPU unitigs, distinct reads found on the device, native sparse flow
        err = None
        try:
            texts = ops.graph_batch(part, names, owned, mine, K, paired, sample, seed, T)
        except Exception as ex:                     # told to every rank below: nobody waits in the gather for a rank that has left
            texts, err = {}, "%s: %s" % (type(ex).__name__, ex)
        if texts is not None:
            lock.release()
            return _gather_and_merge(texts, names, res, gk, group, rank, W, lock, tick, error=err, ops=ops, double_stranded=double_stranded)

    def one(i):
        singles, comps = ops.graph(part, names[i], mine.get(i, []), K, paired)      # pieces: [(global indices, reads)] per source rank
        return (i, singles, comps)

    nthreads = min(len(owned), getattr(ops, "graph_threads", 1))
    if nthreads > 1:                      # owned partitions are independent; the native stage releases the GIL
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            jobs = list(pool.map(one, owned))
    else:
        jobs = [one(i) for i in owned]
    tick("graph", t0)
    t0 = time.time()
    flat = [(c["nodes"], c["edges"], c["paths"]) for _, _, comps in jobs for c in comps]
    ids = [c for _, _, comps in jobs for c in range(len(comps))]
    trs = ops.sparse_flow(flat, ids, seed) if flat else []
    k = 0
    texts = {}
    for i, singles, comps in jobs:
        sname = "%s_%s" % (sample, names[i])
        txt = ""
        for c in range(len(comps)):
            txt += sparse_flow.fasta_records(sname, str(c), trs[k])
            k += 1
        txt += sparse_flow.single_nodes_fasta(sname, singles)
        texts[i] = txt
    tick("sparse flow", t0)
    lock.release()
    return _gather_and_merge(texts, names, res, gk, group, rank, W, lock, tick, ops=ops, double_stranded=double_stranded)


class LocalRows(object):
    """the reads of a partition as doubled read indices into the owner's own resident input (no rows collected)"""
    def __init__(self, sel):
        self.sel = np.ascontiguousarray(sel, dtype=np.uint32)

    def __len__(self):
        return len(self.sel)


class _Texts(Mapping):
    """{partition name: reconstructed FASTA}; the texts stay the byte arrays they arrived as until somebody reads one"""
    def __init__(self, names, arrays):
        self._names, self._a = list(names), dict(zip(names, arrays))

    def __getitem__(self, k):
        return bytes(memoryview(self._a[k])).decode()

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)


def _as_bytes(t):
    return np.frombuffer(t.encode(), dtype=np.uint8) if isinstance(t, str) else np.ascontiguousarray(t, dtype=np.uint8)


def _gather_and_merge(texts, names, res, gk, group, rank, W, lock, tick, error=None, ops=None, double_stranded=True):
    """the per-partition FASTA of every owner to rank 0, which merges (shannon.py:584-604).  error: what went wrong in this rank's
    graph stage, if anything -- it travels with the gather, so every rank raises together and none is left waiting."""
    import time
    t0 = time.time()
    # what went wrong, if anything (a few bytes per rank); then the texts as bytes to rank 0 -- what rank 0 produced itself stays put
    errs = exchange.all_gather_object(error, group, "FASTA gather (status of the ranks)")
    errors = ["rank %d: %s" % (r, e) for r, e in enumerate(errs) if e]
    if errors:
        tick("x:gather fasta", t0)
        raise RuntimeError("graph stage failed on " + "; ".join(errors))
    mine = {int(i): _as_bytes(t) for i, t in texts.items()}
    merged = {}
    if W > 1:
        if rank != 0:
            idx = sorted(mine)
            head = np.asarray([len(idx)] + [v for i in idx for v in (i, mine[i].size)], dtype=np.int64).view(np.uint8)
            out = np.concatenate([head] + [mine[i] for i in idx])
        bufs = [out if (d == 0 and rank != 0) else np.zeros(0, np.uint8) for d in range(W)]
        got = exchange.all_to_all_bytes(bufs, getattr(ops, "device", None), group, "FASTA gather (per-partition transcripts to rank 0)")
        if rank == 0:
            for src, b in enumerate(got):
                if src == 0 or b.size == 0:
                    continue
                n = int(b[:8].view(np.int64)[0])
                head = b[8:8 + 16 * n].view(np.int64).reshape(n, 2)
                pos = 8 + 16 * n
                for i, sz in head.tolist():
                    merged[int(i)] = b[pos:pos + sz]
                    pos += sz
    tick("x:gather fasta", t0)
    if rank != 0:
        return None
    lock.acquire()
    t0 = time.time()
    merged.update(mine)
    single = "".join(">Single_%d\n%s\n" % (i, c) for i, c in enumerate(res.single_contigs))
    order = [merged[i] for i in range(len(names))]
    # the merge over the texts as they are (process_concatenated_fasta.py + faster_reps.py: post.finalize_texts, on the device when
    # the ops have a context); transcripts with characters outside ACGT go through the Python form
    try:
        final = post.finalize_texts([single] + order, double_stranded, ctx=getattr(ops, "ctx", None))
    except _lib.ShannonError as ex:
        if "non-ACGT" not in str(ex) and "empty line" not in str(ex):
            raise
        lines = single.splitlines(True)
        for a in order:
            lines += bytes(memoryview(a)).decode().splitlines(True)
        final = post.finalize(lines, double_stranded)
    parts = _Texts(names, order)
    tick("merge (rank 0)", t0)
    lock.release()
    return {"partitions": parts, "final": final, "contigs": res.contigs,
            "n_k1mers": int(gk.numel()) if gk is not None else int(getattr(res, "n_k1mers_table", 0)),
            "extension": {k: getattr(res, k, None) for k in ("iterations", "n_walks", "total_steps", "wave_steps", "fresh_steps", "dense_rounds", "settled_walks")}}


def _a2a_objects(recv, payload, group):
    """all-to-all of python objects via all_gather_object (small payloads only)."""
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    everything = [None] * W
    dist.all_gather_object(everything, payload, group=group)
    for src in range(W):
        recv[src] = everything[src][rank]


def component_table(cc, group, tick, lock, dev):
    """Component labelling on owner shards, the collectives around it.  cc: this rank's shard behind the steps of
    include/shannon_hip.h's shn_cc_* (GpuOps: the device kernels; the CPU tests: a numpy restatement) --
      cc.n                                     k1-mers of the shard
      cc.queries() -> (keys i64, roots i32, per destination rank)      what higher ranks are asked
      cc.answer(keys, roots, per source rank, base) -> edges i64 [2 E] pairs of global ids (base[r] + local root)
      cc.solve(edges, id_limit) -> (ids ascending, label of each)       the component graph, the same on every rank
      cc.labels(base_me, ids, labels) -> i64 [n]                        global label of every k1-mer of the shard
      cc.sizes(glabel, at_least) -> (labels, counts) numpy              what this rank holds of the larger components
      cc.shard(glabel, big labels ascending, their ranks) -> (keys i64, counts i32, per destination rank)
    Collectives: the shard sizes (all-gather), the queries (all-to-all), the edges (all-gather), the sizes of the large components
    (objects), the pairs of whole components (all-to-all).  Returns ((keys, counts) received: this rank's components, k1-mers of
    the job).  lock: held while this rank computes (see assemble_distributed)."""
    import time
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    lock = lock or _NoLock()
    cdev = exchange.coll_device(dev, group)
    t0 = time.time()
    n_loc = int(cc.n)
    qk, ql, per = cc.queries()
    tick("labels: local components + queries", t0)
    lock.release()
    t0 = time.time()
    nt = torch.tensor([n_loc], dtype=torch.int64, device=cdev)
    nts = [torch.zeros_like(nt) for _ in range(W)]
    dist.all_gather(nts, nt, group=group)
    sizes = [int(x.item()) for x in nts]
    base = [sum(sizes[:r]) for r in range(W)]
    n_glob = sum(sizes)
    rk, rl, rcl = exchange.all_to_all_pairs(qk, ql, per, group, name="component labelling: neighbour queries to the owners (all-to-all)")
    del qk, ql
    tick("x:label queries", t0)
    lock.acquire()
    t0 = time.time()
    edges = cc.answer(rk, rl, rcl, base)
    del rk, rl
    tick("labels: answers", t0)
    lock.release()
    t0 = time.time()
    ge, _ = _all_gather_var(edges, group, name="component labelling: edges between shards (all-gather)")
    del edges
    tick("x:label edges", t0)
    lock.acquire()
    t0 = time.time()
    ids, labels = cc.solve(ge, n_glob + 1)
    del ge
    glabel = cc.labels(base[rank], ids, labels)
    del ids, labels
    # the large components are balanced by size (largest first onto the least loaded rank, as shn_extend_sharded does on a
    # replicated table); everything else goes by the hash of its label.  A component of T k1-mers spread evenly has T / W of them
    # here: every rank reports what it holds of components above a quarter of that, every rank sums the same reports.
    T = max(1024, n_glob // (64 * W))
    mine = cc.sizes(glabel, max(32, T // (4 * W)))
    tick("labels: component graph", t0)
    lock.release()
    t0 = time.time()
    parts = exchange.all_gather_object(mine, group, "component labelling: sizes of the large components")
    tick("x:label sizes", t0)
    lock.acquire()
    t0 = time.time()
    tot = {}
    for lab, cnt in parts:
        for a, b in zip(np.asarray(lab).tolist(), np.asarray(cnt).tolist()):
            tot[a] = tot.get(a, 0) + b
    big = sorted((a for a, b in tot.items() if b >= T // 2), key=lambda a: (-tot[a], a))
    load = [0] * W
    own = {}
    for a in big:
        r = min(range(W), key=lambda q: (load[q], q))
        own[a] = r
        load[r] += tot[a]
    bl = sorted(own)
    sk, sc, send = cc.shard(glabel, np.asarray(bl, dtype=np.int64), np.asarray([own[a] for a in bl], dtype=np.uint8))
    del glabel
    tick("labels: owners + shard", t0)
    lock.release()
    t0 = time.time()
    rk, rc, _ = exchange.all_to_all_pairs(sk, sc, send, group, name="component exchange (the k1-mers of whole components to their rank, all-to-all)")
    del sk, sc
    tick("x:component exchange", t0)
    lock.acquire()
    return (rk, rc), n_glob


class _GpuComponents(object):
    """the steps of component_table() on the device (device.ComponentShards = shn_cc_*), tensors on the rank's GPU"""

    def __init__(self, ops, owned, W, rank):
        self.ops, self.dev, self.n = ops, ops.device, len(owned)
        self.cc = ops._dev.ComponentShards(ops.ctx, owned, W, rank)

    def queries(self):
        per = self.cc.query_counts()
        nq = int(per.sum())
        qk = torch.empty(max(nq, 1), dtype=torch.int64, device=self.dev)
        ql = torch.empty(max(nq, 1), dtype=torch.int32, device=self.dev)
        self.cc.queries(qk.data_ptr(), ql.data_ptr())
        return qk, ql, per

    def answer(self, rk, rl, rcl, base):
        torch.cuda.synchronize()
        edges = torch.empty(2 * max(int(sum(rcl)), 1), dtype=torch.int64, device=self.dev)
        ne = self.cc.answer(rk.data_ptr(), rl.data_ptr(), rcl, base, edges.data_ptr())
        return edges[:2 * ne]

    def solve(self, ge, id_limit):
        torch.cuda.synchronize()
        E = int(ge.numel()) // 2
        ids = torch.empty(2 * max(E, 1), dtype=torch.int64, device=self.dev)
        labels = torch.empty(2 * max(E, 1), dtype=torch.int64, device=self.dev)
        nn = self.cc.solve(self.ops.ctx, ge.data_ptr(), E, id_limit, ids.data_ptr(), labels.data_ptr())
        return ids[:nn], labels[:nn]

    def labels(self, base_me, ids, labels):
        glabel = torch.empty(max(self.n, 1), dtype=torch.int64, device=self.dev)
        self.cc.labels(base_me, ids.data_ptr(), labels.data_ptr(), int(ids.numel()), glabel.data_ptr())
        return glabel

    def sizes(self, glabel, at_least):
        torch.cuda.synchronize()
        u, c = torch.unique(glabel[:self.n], return_counts=True)
        sel = c >= at_least
        return u[sel].cpu().numpy(), c[sel].cpu().numpy()

    def shard(self, glabel, big, big_owner):
        n_big = len(big)
        dbig = torch.as_tensor(big if n_big else np.zeros(1, np.int64), device=self.dev)
        dbo = torch.as_tensor(big_owner if n_big else np.zeros(1, np.uint8), device=self.dev)
        owner = torch.empty(max(self.n, 1), dtype=torch.uint8, device=self.dev)
        torch.cuda.synchronize()
        self.cc.owners(glabel.data_ptr(), dbig.data_ptr(), dbo.data_ptr(), n_big, owner.data_ptr())
        sk = torch.empty(max(self.n, 1), dtype=torch.int64, device=self.dev)
        sc = torch.empty(max(self.n, 1), dtype=torch.int32, device=self.dev)
        send = self.cc.shard(owner.data_ptr(), sk.data_ptr(), sc.data_ptr())
        return sk, sc, send

    def close(self):
        self.cc.close()


class GpuOps(object):
    """Per-rank compute on the local GPU through the C ABI."""

    def __init__(self, ctx, d1, d2, store, K):
        from . import device
        self.ctx, self.d1, self.d2, self.store, self.K = ctx, d1, d2, store, K
        self.paired = d2 is not None
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._dev = device

    supports_strand_specific = True
    strand_specific = False              # set by assemble_distributed
    min_weight, min_length, kmer_hard_cutoff = 3, 75, 1      # set by assemble_distributed (shannon.py:55-57, 237-247)

    def n_reads(self):
        return len(self.d1)

    def _hard_cutoff(self, t):
        """`jellyfish dump -L` (shannon.py:441) on a table of REDUCED counts (the job's, not a rank's slice)"""
        if self.kmer_hard_cutoff <= 1:
            return t
        kept = t.filter_lower(self.kmer_hard_cutoff)
        t.close()
        return kept

    def _count(self):
        if self.strand_specific:         # forward counting of reads_1 and RC(reads_2): a table of plain (not canonical) k1-mers
            return self._dev.count_k1mers_strand_specific(self.ctx, self.d1, self.d2, self.K + 1)
        return self._dev.count_k1mers(self.ctx, [self.d1, self.d2] if self.paired else [self.d1], self.K + 1, True)

    @property
    def owner_labelling(self):
        """components labelled on the owner shards, no replicated table (SHN_OWNER_LABELS=0: the table all-gathered to every rank and
        labelled there -- the path of rounds 2-4, kept for comparison)"""
        import os
        return os.environ.get("SHN_OWNER_LABELS", "1") != "0"

    def _digest(self, name, table):
        """SHN_DIST_DIGEST=1 (diagnostics): a checksum of a table's sorted (key, count) pairs per stage, in self.digests"""
        import os
        if os.environ.get("SHN_DIST_DIGEST") != "1":
            return
        import hashlib
        k, c = table.download()
        o = np.argsort(k, kind="stable")
        h = hashlib.sha256(np.ascontiguousarray(k[o]).tobytes() + np.ascontiguousarray(c[o]).tobytes()).hexdigest()[:16]
        if not hasattr(self, "digests"):
            self.digests = {}
        self.digests[name] = (int(len(k)), h)

    def local_pairs(self, W, by_minimizer=False):
        t = self._count()
        self._digest("counted", t)
        n = len(t)
        dk = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        dc = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        per = (t.shard_by_minimizer if by_minimizer else t.shard)(W, dk.data_ptr(), dc.data_ptr())
        t.close()
        return dk, dc, per

    def owned_table(self, rk, rc):
        """the pairs this rank owns, equal keys summed"""
        torch.cuda.synchronize()
        t = self._hard_cutoff(self._dev.Table.from_pairs(self.ctx, rk.data_ptr(), rc.data_ptr(), rk.numel(), self.K + 1, not self.strand_specific))
        self._digest("owned", t)
        return t

    def component_table(self, owned, group, tick):
        """owned: this rank's shard of the k1-mers (by minimizer).  Returns (a table of the whole components dealt to this rank,
        the number of k1-mers of the job): component_table() below with the device kernels (include/shannon_hip.h: shn_cc_*)."""
        import time
        W, rank = dist.get_world_size(group), dist.get_rank(group)
        n_loc = len(owned)
        t0 = time.time()
        cc = _GpuComponents(self, owned, W, rank)
        tick("labels: local components + queries", t0)
        try:
            pairs, n_glob = component_table(cc, group, tick, getattr(self, "lock", None), self.device)
        finally:
            cc.close()
        t0 = time.time()
        owned.close()
        rk, rc = pairs
        torch.cuda.synchronize()
        table = self._dev.Table.from_pairs(self.ctx, rk.data_ptr(), rc.data_ptr(), rk.numel(), self.K + 1, not self.strand_specific)
        tick("table", t0)
        self._digest("components", table)
        self.component_table_sizes = (int(n_loc), int(len(table)), int(n_glob))      # owned shard, walked table, the job (tests, bench)
        return table, n_glob

    def local_table(self):
        """one-rank job: the counted table itself (no export to pairs)"""
        return self._hard_cutoff(self._count())

    def reduce_pairs(self, rk, rc):
        torch.cuda.synchronize()
        t = self._hard_cutoff(self._dev.Table.from_pairs(self.ctx, rk.data_ptr(), rc.data_ptr(), rk.numel(), self.K + 1, not self.strand_specific))
        n = len(t)
        dk = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        dc = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        t.shard(1, dk.data_ptr(), dc.data_ptr())           # the reduced pairs stay on the device (a one-way "shard")
        t.close()
        return dk[:n], dc[:n]

    def table_from_pairs(self, gk, gc):
        torch.cuda.synchronize()
        t = self._dev.Table.from_pairs(self.ctx, gk.data_ptr(), gc.data_ptr(), gk.numel(), self.K + 1, not self.strand_specific)
        self._digest("gathered", t)
        return t

    graph_threads = 8
    sharded_extension = True

    @property
    def resident_rows(self):
        """a partition whose routed reads are all rows of this rank's resident input is handed to the graph stage by rows"""
        def _matrix(m):
            return isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.ndim == 2 and m.flags["C_CONTIGUOUS"]
        import os
        return (self.unitigs is not None and _matrix(getattr(self.store, "r1", None)) and (not self.paired or _matrix(getattr(self.store, "r2", None)))
                and os.environ.get("SHN_GRAPH_ROWS", "1") != "0")
    array_payload = True               # collect() returns (code rows, strand flags): travels as bytes, not as pickles

    def extension(self, table, partition_size, group=None, presharded=None):
        """replicated table -> walks AND contig stages sharded by connected component of the k1-mer graph; the accepted
        contigs + their connections (a few MB) are all-gathered and merged in the global walk order.
        presharded = the number of k1-mers of the job: `table` holds the whole components dealt to this rank already (component_table);
        the walks are the unsharded ones on it, everything after them as before."""
        import time
        from . import extension_correction as ec
        W, rank = dist.get_world_size(group), dist.get_rank(group)
        lock = getattr(self, "lock", None) or _NoLock()
        self.coll_wait = 0.0               # seconds inside the contig stage's collectives (waiting for the slowest rank included)

        class Gather(object):                     # the collectives of the sharded contig stage
            world, rank = W, None

            @staticmethod
            def all_gather(obj):
                lock.release()
                t0 = time.time()
                parts = exchange.all_gather_object(obj, group, "contig gathers (accepted contigs + connections of the shards)")
                self.coll_wait += time.time() - t0
                lock.acquire()
                return parts

            @staticmethod
            def all_gather_arrays(arrays):
                lock.release()
                t0 = time.time()
                parts = exchange.all_gather_arrays(arrays, exchange.coll_device(self.device, group), group,
                                                   "contig gathers (candidates of the shards: seed weights, keys, offsets, text)")
                self.coll_wait += time.time() - t0
                lock.acquire()
                return parts

            @staticmethod
            def all_reduce_max(v):
                t = torch.tensor([int(v)], dtype=torch.int64, device=exchange.coll_device(self.device, group))
                lock.release()
                t0 = time.time()
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                self.coll_wait += time.time() - t0
                lock.acquire()
                return int(t.item())

        Gather.rank = rank
        # (a large, diverse table -- BASELINE configs[2] and beyond: 10^8+ k1-mers, 10^5+ candidate contigs -- takes the path
        # "sharded walks + one replicated GPU contig stage" inside run_correction; a few deep gene families per rank -- the
        # weak-scaling workload, 5 M k1-mers per family -- keep their contig stages sharded too: SHN_CONTIG_GPU=1 / 0 forces either)
        res = ec.run_correction(self.ctx, table, self.min_weight, self.min_length, partition_size, want_allowed=False, shard=None if presharded is not None else (W, rank),
                                gather=Gather, timings=getattr(self, "timings", None), table_size=presharded)
        table.close()
        if presharded is not None:
            res.n_k1mers_table = int(presharded)
        return res

    def route(self, res, K, partition_size, part_vectors):
        from . import kmers_for_component as kfc, mbgraph_native
        import os
        self.unitigs, self.part_index = None, None
        gpu_graph = K <= 31 and os.environ.get("SHN_GRAPH_GPU", "1") != "0"
        part = kfc.kmers_for_component(self.ctx, res, self.d1, self.d2, K, partition_size, part_vectors=part_vectors, want_rows=False,
                                       timings=getattr(self, "timings", None), lazy_routes=True, lazy_graph_inputs=gpu_graph,
                                       strand_specific=self.strand_specific)
        names = list(part["new_components"])
        if gpu_graph and names:
            # the raw K-mer graphs of ALL partitions contracted on every rank (a tenth of a second at 111 partitions): every rank
            # needs every partition's K-mer count for the read caps, the owners need the unitigs
            self.unitigs = mbgraph_native.Unitigs(self.ctx, [part["new_components"][nm] for nm in names], K, flat_text=part.get("flat_text"))
            self.part_index = {nm: i for i, nm in enumerate(names)}
        return part

    def n_nodes(self, part, name, K):
        if self.unitigs is not None:
            return self.unitigs.n_kmers(self.part_index[name])
        return part["n_kmer_nodes"][name]

    @staticmethod
    def _merge_pieces(pieces):
        """the pieces of one partition from their source ranks -> (code rows, strand flags) in the global strand-doubled order"""
        pieces = [p for p in pieces if len(p[0])]
        if not pieces:
            return np.zeros((0, 1), np.uint8), np.zeros(0, np.uint8)
        one = len(pieces) == 1                                    # (no copy of a 100 MB piece)
        gidx = pieces[0][0] if one else np.concatenate([g for g, _ in pieces])
        rows = np.ascontiguousarray(pieces[0][1][0] if one else np.concatenate([d[0] for _, d in pieces]))
        rc1 = np.ascontiguousarray(pieces[0][1][1] if one else np.concatenate([d[1] for _, d in pieces]))
        if len(gidx) > 1 and not bool((gidx[1:] > gidx[:-1]).all()):      # (one source rank: already in order)
            from . import _lib
            order = np.argsort(gidx, kind="stable")
            rows = _lib.gather_rows(rows, order)
            rc1 = np.ascontiguousarray(rc1[order])
        return rows, rc1

    def graph_batch(self, part, names, owned, mine, K, paired, sample, seed, T):
        """multibridging.main + algorithm_SF for the owned partitions, as the single-GPU pipeline runs them: the received code rows
        become a resident read set of their own (row i forward = doubled index i, reverse strand = n + i), the distinct reads are
        found on the device (shn_mbgraph_run_rows), all components go through shn_sparse_flow in one call.  {partition: FASTA}."""
        import time, threading
        from concurrent.futures import ThreadPoolExecutor
        from . import mbgraph_native, _lib
        if self.unitigs is None:
            return None
        t0 = time.time()
        up = threading.Lock()

        def one(i):
            pieces = mine.get(i, [])
            local = pieces[0][1] if (len(pieces) == 1 and isinstance(pieces[0][1], LocalRows)) else None
            rows, rc1 = (np.zeros((0, 1), np.uint8), np.zeros(0, np.uint8)) if local is not None else self._merge_pieces(pieces)
            rb = None
            for _attempt in (0, 1):
                try:
                    if self.strand_specific and len(rows):
                        # -s: the rows hold reads_1[i] and, for pairs, reads_2[i] side by side (collect); the second mate is read on its
                        # reverse strand (rc2 = 1): the pairs (reads_1[i], RC(reads_2[i])) of shannon.py:407-411, not strand-doubled
                        n, L2 = rows.shape
                        L = L2 // 2 if paired else L2
                        b1 = np.ascontiguousarray(rows[:, :L]).reshape(-1)
                        o1 = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
                        b2 = np.ascontiguousarray(rows[:, L:]).reshape(-1) if paired else None
                        return mbgraph_native.run_partition_handle(rb, 0 if rb is None else len(rb) // (K + 1), K, b1, o1, b2, o1 if paired else None,
                                                                   ctx=self.ctx, enc=1, rc1=np.zeros(n, np.uint8), rc2=np.ones(n, np.uint8) if paired else None,
                                                                   unitigs=self.unitigs, part=i)
                    if local is not None and len(local):
                        # this rank's own reads, named by their rows in its resident input (as pipeline.assemble_resident does)
                        return mbgraph_native.run_partition_rows(self.ctx, self.unitigs, i, self.d1, self.d2 if paired else None, self.store.r1,
                                                                 self.store.r2 if paired else None, local.sel, rb, 0 if rb is None else len(rb) // (K + 1))
                    if len(rows) == 0:
                        z, o = np.zeros(1, np.uint8), np.zeros(1, np.uint64)
                        return mbgraph_native.run_partition_handle(rb, 0 if rb is None else len(rb) // (K + 1), K, z, o, z if paired else None,
                                                                   o if paired else None, ctx=self.ctx, unitigs=self.unitigs, part=i)
                    with up:
                        d = self._dev.Reads.from_codes(self.ctx, rows)
                    n = len(rows)
                    didx = (np.arange(n, dtype=np.int64) + np.where(rc1 != 0, n, 0)).astype(np.uint32)
                    try:
                        return mbgraph_native.run_partition_rows(self.ctx, self.unitigs, i, d, d if paired else None, rows, rows if paired else None,
                                                                 didx, rb, 0 if rb is None else len(rb) // (K + 1))
                    finally:
                        d.close()
                except _lib.ShannonError as ex:
                    if rb is not None or "needs the k1-mer rows" not in str(ex):
                        raise
                    rb_ = part["k1mer_bytes"][names[i]]                  # a partition holding a cycle of condensable edges
                    rb = rb_() if callable(rb_) else rb_
        nthreads = max(1, min(len(owned), _lib.host_cpus()))
        graphs, futs = [], []
        try:
            if nthreads > 1:
                with ThreadPoolExecutor(max_workers=nthreads) as pool:
                    futs = [pool.submit(one, i) for i in owned]
                    graphs = [f.result() for f in futs]
            else:
                for i in owned:
                    graphs.append(one(i))
            T["graph"] = T.get("graph", 0.0) + time.time() - t0
            t0 = time.time()
            texts = mbgraph_native.sparse_flow_native(self.ctx, graphs, ["%s_%s" % (sample, names[i]) for i in owned], seed, raw=True) if owned else []
            T["sparse flow"] = T.get("sparse flow", 0.0) + time.time() - t0
            return {i: txt for i, txt in zip(owned, texts)}
        finally:
            # whatever happened: the graphs built so far and the unitigs go back now (a failing partition must not leave them to
            # the garbage collector; the caller tells the other ranks before the gather, _gather_and_merge)
            for g in (graphs or [f.result() for f in futs if f.done() and not f.cancelled() and f.exception() is None]):
                g.close()
            self.unitigs.close()
            self.unitigs = None

    def collect(self, sel):
        """the reads of the doubled indices `sel` as they travel to a partition's owner: stored code rows + strand
        flags of the first mates (second mates: same rows, opposite strand)"""
        if self.strand_specific:
            # plain read indices: reads_1[i] as stored and, for pairs, reads_2[i] as stored beside it (the owner reads it reversed)
            b1, o1, _rc, enc = self.store.gather_codes_ss(sel, 1)
            if enc != 1:
                raise ValueError("GpuOps: -s on the N-rank path needs the reads stored as code matrices")
            L = int(o1[1] - o1[0]) if len(o1) > 1 else 0
            rows = b1.reshape(len(sel), L)
            if self.paired:
                b2, o2, _rc2, _e = self.store.gather_codes_ss(sel, 2)
                rows = np.concatenate([rows, b2.reshape(len(sel), L)], axis=1)
            return (np.ascontiguousarray(rows), np.zeros(len(sel), np.uint8))
        buf, off, rc1, enc = self.store.gather_codes(sel, 1)
        L = int(off[1] - off[0]) if len(off) > 1 else 0
        return (buf.reshape(len(sel), L), rc1)

    def graph(self, part, name, pieces, K, paired):
        from . import mbgraph_native
        rb = part["k1mer_bytes"][name]                     # the partition's k1-mer file as fixed-width rows
        if any(len(p[0]) for p in pieces):
            # global strand-doubled order = the reference's file order
            pieces = [p for p in pieces if len(p[0])]
            one = len(pieces) == 1                                    # (no copy of a 100 MB piece)
            gidx = pieces[0][0] if one else np.concatenate([g for g, _ in pieces])
            rows = np.ascontiguousarray(pieces[0][1][0] if one else np.concatenate([d[0] for _, d in pieces]))
            rc1 = np.ascontiguousarray(pieces[0][1][1] if one else np.concatenate([d[1] for _, d in pieces]))
            if len(gidx) > 1 and not bool((gidx[1:] > gidx[:-1]).all()):      # (one source rank: already in order)
                from . import _lib
                order = np.argsort(gidx, kind="stable")
                rows = _lib.gather_rows(rows, order)
                rc1 = np.ascontiguousarray(rc1[order])
        else:
            rows, rc1 = np.zeros((0, 1), np.uint8), np.zeros(0, np.uint8)
        if self.strand_specific and len(rows):
            # -s: reads_1[i] | reads_2[i] side by side (collect), the second mate read on its reverse strand
            n, L2 = rows.shape
            L = L2 // 2 if paired else L2
            off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
            b1 = np.ascontiguousarray(rows[:, :L]).reshape(-1)
            b2 = np.ascontiguousarray(rows[:, L:]).reshape(-1) if paired else None
            singles, comps, _log = mbgraph_native.run_partition_arrays(rb if len(rb) else np.zeros(1, np.uint8), len(rb) // (K + 1), K, b1, off, b2,
                                                                       off if paired else None, ctx=self.ctx, enc=1, rc1=np.zeros(n, np.uint8),
                                                                       rc2=np.ones(n, np.uint8) if paired else None)
            return singles, comps
        off = np.arange(len(rows) + 1, dtype=np.uint64) * np.uint64(rows.shape[1])
        b1 = rows.reshape(-1) if rows.size else np.zeros(1, np.uint8)
        singles, comps, _log = mbgraph_native.run_partition_arrays(rb if len(rb) else np.zeros(1, np.uint8), len(rb) // (K + 1), K,
                                                                   b1, off, b1 if paired else None, off if paired else None, ctx=self.ctx,
                                                                   enc=1, rc1=rc1, rc2=(1 - rc1).astype(np.uint8) if paired else None)
        return singles, comps

    def sparse_flow(self, flat, ids, seed):
        from .pipeline import _sparse_flow_with_ids
        return _sparse_flow_with_ids(self.ctx, flat, ids, seed)
