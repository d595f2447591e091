"""End-to-end hot path in memory: reads -> (K+1)-mer table -> contigs/partitions -> per-partition
multibridged graph -> sparse-flow transcripts -> merged FASTA.  Mirrors the order of shannon.py
(394-647) and run_MB_SF_fn.py (210-254) without the file round trips between stages (the
reference's --inMem hand-off); `shannon.py` at the repo root adds the CLI and the OUT/ file tree.
"""
import os
import time
import numpy as np
from . import _lib, device, extension_correction as ec, kmers_for_component as kfc, mbgraph, mbgraph_native, sparse_flow, post


class Result(object):
    """.partitions {name: record}, .final {name: sequence}, .all_reconstructed (the lines of all_reconstructed.fasta; joined from
    the partitions' texts on first use), .extension, .timings ..."""
    _texts = None

    @property
    def all_reconstructed(self):
        if self.__dict__.get("_lines") is None and self._texts is not None:
            lines = []
            for t in self._texts:
                lines += (t if isinstance(t, str) else bytes(t).decode()).splitlines(True)
            self.__dict__["_lines"] = lines
        return self.__dict__.get("_lines")

    @all_reconstructed.setter
    def all_reconstructed(self, lines):
        self.__dict__["_lines"] = lines


class PartitionRecord(dict):
    """Record of one partition: n_reads_routed, n_k1mers, reconstructed_fasta as plain entries; `singles`, `components`
    and `log` (the tables of mbgraph.output_components) are exported from the native graph object on first access."""

    def __init__(self, n_reads_routed, n_k1mers, graph):
        dict.__init__(self, n_reads_routed=n_reads_routed, n_k1mers=n_k1mers)
        self.graph = graph

    def __missing__(self, key):
        if key == "reconstructed_fasta" and getattr(self, "fasta_raw", None) is not None:
            dict.__setitem__(self, key, bytes(self.fasta_raw).decode())
            return dict.__getitem__(self, key)
        if key in ("singles", "components", "log") and self.graph is not None:
            singles, comps, log = self.graph.tables()
            dict.update(self, singles=singles, components=comps, log=log)
            return dict.__getitem__(self, key)
        raise KeyError(key)


_POOLS = {}


def _graph_pool(n_threads):
    """the host threads of the graph stage, kept from call to call: every thread owns a stream and device workspaces of its own
    (shn_thread_ctx) that go with it -- a fresh pool per step created and freed them sixteen times a step"""
    from concurrent.futures import ThreadPoolExecutor
    pool = _POOLS.get(n_threads)
    if pool is None:
        pool = _POOLS[n_threads] = ThreadPoolExecutor(max_workers=n_threads, thread_name_prefix="shn-graph")
    return pool


def n_kmer_nodes(rows, K):
    """#distinct K-mers among the k1-mer rows = len(Node.nodes) after loading (multibridging.py:383)."""
    s = set()
    for km, _ in rows:
        s.add(km[:-1])
        s.add(km[1:])
    return len(s)


def assemble(ctx, reads1, reads2=None, K=25, partition_size=500, min_weight=3, min_length=75, overload=2, penalty=5,
             sample="shannon", seed=0, double_stranded=True, part_vectors=None, timings=None, hits_factory=None,
             native_graph=True, kmer_hard_cutoff=1):
    """reads1/reads2: lists of strings or uint8 code matrices (reads2 None = single-end).
    min_weight = the reference's hyp_min_weight (--kmer_soft_cutoff, shannon.py:243-247, 457); kmer_hard_cutoff = its
    jellyfish_kmer_cutoff (--kmer_hard_cutoff, `jellyfish dump -L`, shannon.py:237-241, 441).
    Returns Result with .partitions {name: dict}, .all_reconstructed (lines), .final {name: seq}."""
    T = timings if timings is not None else {}
    paired = reads2 is not None

    def tick(name, t0):
        T[name] = T.get(name, 0.0) + time.time() - t0

    t0 = time.time()
    mk = (lambda r: device.Reads.from_strings(ctx, r)) if isinstance(reads1[0], str) else (lambda r: device.Reads.from_codes(ctx, r))
    d1 = mk(reads1)
    d2 = mk(reads2) if paired else None
    store = kfc.ReadStore(reads1, reads2)
    tick("upload+pack", t0)
    return assemble_resident(ctx, d1, d2, store, K, partition_size, min_weight, min_length, overload, penalty, sample, seed,
                             double_stranded, part_vectors, T, hits_factory, native_graph, kmer_hard_cutoff=kmer_hard_cutoff)


def assemble_resident(ctx, d1, d2, store, K=25, partition_size=500, min_weight=3, min_length=75, overload=2, penalty=5,
                      sample="shannon", seed=0, double_stranded=True, part_vectors=None, timings=None, hits_factory=None,
                      native_graph=True, graph_threads=None, keep_partitioning=False, defer_back=False, kmer_hard_cutoff=1):
    """Same as assemble() with the reads already packed in HBM (d1/d2: device.Reads).  graph_threads: partitions whose
    graph stage may run concurrently on host threads.  keep_partitioning: leave the partition stage's tables (partition ->
    contigs, routed read indices) on the result as `.partitioning` (tests/test_fullsize_gpu.py reads them).  defer_back: run count,
    extension, partitioning / routing and the unitig batch now and return a function that does the rest (see `back`);
    defer_back="early": only count and extension now, partitioning / routing / unitigs with the rest (see `middle`)."""
    # double_stranded=False: -s / --ss / --strand_specific.  shannon.py:394-424 then leaves single-end reads as they are and
    # reverse-complements the second mates, without doubling; from :427 on double_stranded is False in BOTH modes, so only the read
    # set differs: forward counting (d2: its reverse complements), routes of plain read indices, pairs (R1[i], RC(R2[i])) in the
    # graph stage; process_concatenated_fasta (:596) gets the user's flag.
    ss = not double_stranded
    T = timings if timings is not None else {}
    paired = d2 is not None
    if graph_threads is None:                   # a rank's share of the host cores (8 ranks per node), at least 8
        import os
        # partition threads: all the CPUs a small allowance grants (a cgroup quota of 16 CPUs: 16), a quarter of a whole machine
        cpus = _lib.host_cpus()
        graph_threads = int(os.environ.get("SHN_GRAPH_THREADS", 0)) or (max(4, cpus) if cpus <= 32 else min(64, cpus // 4))
    if hits_factory is None:
        from . import graph_seeds
        hits_factory = graph_seeds.hits_factory(ctx)

    def tick(name, t0):
        T[name] = T.get(name, 0.0) + time.time() - t0

    R = Result()
    t0 = time.time()
    table = (device.count_k1mers_strand_specific(ctx, d1, d2, K + 1) if ss else
             device.count_k1mers(ctx, [d1, d2] if paired else [d1], K + 1, both_strands=True))
    if kmer_hard_cutoff > 1:
        # `jellyfish dump -L N` (shannon.py:441): k1-mers counted fewer than N times never reach k1mer.dict_org -- no seed, no
        # extension step, no weight of a later stage sees them (the seed threshold min_weight is another matter: a k1-mer below
        # THAT is still walked over, extension_correction.py:229-233)
        kept = table.filter_lower(kmer_hard_cutoff)
        table.close()
        table = kept
    R.n_k1mers, R.n_windows = len(table), table.total
    tick("count", t0)
    t0 = time.time()
    res = ec.run_correction(ctx, table, min_weight, min_length, partition_size, want_allowed=not native_graph, timings=T)
    table.close()
    R.extension = res
    tick("extension", t0)
    part = names = unitigs = single_text = sf_jobs = gpu_unitigs = None

    def middle(mctx):
        """partition + route + the unitig batch, on context mctx: the caller's, or -- defer_back="early" -- the back half's (the
        extension's result is on the host or in device buffers whose producers have been waited for by then).  These stages use the
        stage workspaces of mctx: beside another context's counting / extension only on a context with a set of its own
        (device.Context(own_workspaces=True))."""
        nonlocal part, names, unitigs, single_text, sf_jobs, gpu_unitigs
        t0 = time.time()
        gpu_unitigs = native_graph and K <= 31 and os.environ.get("SHN_GRAPH_GPU", "1") != "0"
        # (reads kept as code matrices + GPU unitigs: the graph stage names a partition's reads by their place in the routes on the
        # device, rows mode below -- the routed lists are then fetched per partition, only by the forms that want them on the host)
        def _is_matrix(m):
            return isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.ndim == 2 and m.flags["C_CONTIGUOUS"]
        rows_likely = (gpu_unitigs and d1 is not None and _is_matrix(getattr(store, "r1", None)) and (not paired or (d2 is not None and _is_matrix(getattr(store, "r2", None))))
                       and os.environ.get("SHN_GRAPH_ROWS", "1") != "0" and not ss)
        part = kfc.kmers_for_component(mctx, res, d1, d2, K, partition_size, overload, penalty, True, part_vectors,
                                       want_rows=not native_graph, timings=T, lazy_graph_inputs=gpu_unitigs, strand_specific=ss,
                                       lazy_routes=rows_likely)
        tick("partition+route", t0)
        if keep_partitioning:
            R.partitioning = part
        R.partitions = {}
        # reconstructed_single_contigs.fasta: one text (the native merge takes texts); its lines only for the Python forms of the back half
        raw, ids = getattr(res, "contig_raw", None), getattr(res, "single_ids", None)
        if raw is not None and ids is not None and len(ids) == len(res.single_contigs) and len(raw[2]) == len(res.contigs):
            single_text = _lib.fasta_records(raw[0], raw[1], raw[2][np.asarray(ids, dtype=np.int64)], "Single_")      # (bytes, from the contig stage's buffer)
        else:
            single_text = "".join([">Single_%d\n%s\n" % (i, c) for i, c in enumerate(res.single_contigs)])
        sf_jobs = []

        names = list(part["new_components"])
        unitigs = None
        if gpu_unitigs and names:
            # the raw K-mer graphs of all partitions, contracted to unitigs in one batch on the GPU
            t0 = time.time()
            unitigs = mbgraph_native.Unitigs(mctx, [part["new_components"][nm] for nm in names], K, flat_text=part.get("flat_text"))
            tick("graph unitigs (GPU)", t0)

    def back(bctx=None):
        """the host-bound half of the step -- graph stage, sparse flow, merge -- on context bctx (default: the caller's): with
        defer_back the caller may run it on another thread and context while it starts the next batch's counting / extension
        (bench.py --overlap); everything it touches on the device from then on is its own (forked graph contexts, LP batches)"""
        ctx_b = bctx if bctx is not None else ctx
        part_index = {nm: i for i, nm in enumerate(names)}
        check_rows = os.environ.get("SHN_GRAPH_CHECK")

        timeline = {} if os.environ.get("SHN_DEBUG_PARTS") else None

        def _matrix(m):
            return isinstance(m, np.ndarray) and m.dtype == np.uint8 and m.ndim == 2 and m.flags["C_CONTIGUOUS"]
        # reads kept as code matrices + GPU unitigs: partitions name their reads by rows (SHN_GRAPH_ROWS=0: gather them on the host)
        # (strand-specific runs hand the reads over as gathered rows + strand flags: the rows mode knows the doubled layout only)
        rows_mode = (unitigs is not None and d1 is not None and _matrix(store.r1) and (not paired or _matrix(store.r2)) and
                     (not paired or d2 is not None) and os.environ.get("SHN_GRAPH_ROWS", "1") != "0" and not ss)

        # the sparse flow of a partition right behind its graph, on the same thread (one algorithm_SF.py process per component in
        # the reference, run_MB_SF_fn.py:242-250): the small partitions are through theirs while the largest is still at its graph
        sflow_beside = (native_graph and len(names) > 1 and graph_threads > 1 and os.environ.get("SHN_SFLOW_NATIVE", "1") != "0" and
                        os.environ.get("SHN_SFLOW_BESIDE", "1") != "0")

        # (the few partitions with the most reads -- their graphs take longest -- each run their own sparse flow; all the others go
        # through ONE call, made by the thread that finishes the last of their graphs: a call's cost is its dependent LP rounds, and
        # a hundred small calls keep sixteen threads waiting on a hundred times as many tiny batches)
        import threading
        n_routed = {nm: len(part["routes"][nm]) for nm in names}
        big_cut = max(n_routed.values()) // 4 if names else 0
        # (at most four: when the partitions are of one size -- a component cut by gpmetis into a hundred parts -- "a quarter of the
        # largest" holds for all of them, and four hundred calls of their own were 225,000 tiny LP batches per step at --config 2p)
        big = set(sorted((nm for nm in names if n_routed[nm] >= max(1, big_cut)), key=lambda nm: -n_routed[nm])[:4]) if sflow_beside else set()
        small_lock, small_done, small_recs = threading.Lock(), [0], {}
        n_small = len(names) - len(big)
        # ... in WAVES: the thread that finishes a graph takes every small partition that is ready and in no wave yet, once there are
        # `wave_min` of them (and fewer than `waves_max` waves are running) or the last graph is done.  With ONE call behind the last graph a
        # hundred partitions of one size (bench.py --config 2p: 401 parts of two gpmetis components) left all their sparse flow --
        # 890 dependent LP rounds, 1.3 s -- to be done after the graph stage with fifteen threads idle; a wave's rounds wait for the
        # device, not for a core, so they run beside the other threads' graphs.  (A component's answer does not depend on its call:
        # its random costs are numbered inside its own graph.)  A tenth of the partitions per wave, four waves in flight: a wave lasts as
        # long as its ~800 dependent rounds whatever its size, so what counts is how little is left for the wave behind the LAST graph
        # (tools/sflow_waves_r06.sh at 2p, per step: a sixth / 3 in flight 4.49-4.61 s, 40 / 4: 4.31-4.34, 32 / 5: 4.37-4.45, 100 / 3: 4.71).
        wave_min = max(16, int(os.environ.get("SHN_SFLOW_WAVE", 0)) or (n_small + 9) // 10)
        waves_max = max(1, int(os.environ.get("SHN_SFLOW_WAVES_MAX", 0)) or 4)
        ready, waves_running = [], [0]
        # ... and the merge takes every text as a piece the moment it exists (post.PostStream: lines, upload, fingerprints beside the
        # graph stage; the order-dependent rules at the end).  Piece 0 = the single contigs, piece 1 + i = partition i.
        pstream, ps_failed, post_futs = None, [], []
        if (sflow_beside and os.environ.get("SHN_POST_NATIVE", "1") != "0" and os.environ.get("SHN_POST_GPU", "1") != "0" and
                os.environ.get("SHN_POST_STREAM", "1") != "0"):
            try:
                # room for the merge's text on the device: the transcripts are paths through the contigs' graph -- a few times the
                # accepted contigs' text (both strands, isoforms sharing exons); a text that outgrows it falls back to the one-piece merge
                _raw = getattr(res, "contig_raw", None)
                _cbytes = int(len(_raw[0])) if _raw is not None else sum(len(c) for c in res.contigs)
                pstream = post.PostStream(ctx_b, capacity=min(2 << 30, max(64 << 20, 8 * _cbytes + len(single_text))))
            except _lib.ShannonError:
                pstream = None

        def post_piece(index, text):
            if pstream is None or ps_failed:
                return
            try:
                pstream.add(index, text)
            except _lib.ShannonError as ex:                   # (no room, a last line without its end, ...: the merge in one piece below)
                ps_failed.append(str(ex))

        def one_partition(name):
            """multibridged graph of one partition (multibridging.main for `name`); returns its record + timings"""
            rec, tt = _one_partition(name)
            if sflow_beside and getattr(rec, "graph", None) is not None:
                t1 = time.time()
                if name in big:
                    rec.fasta_raw = mbgraph_native.sparse_flow_native(ctx_b, [rec.graph], ["%s_%s" % (sample, name)], seed, raw=True, threaded=True)[0]
                    post_piece(1 + part_index[name], rec.fasta_raw)
                else:
                    with small_lock:
                        small_recs[name] = rec
                        ready.append(name)
                        small_done[0] += 1
                        last = small_done[0] == n_small
                        wave = []
                        if last or (len(ready) >= wave_min and waves_running[0] < waves_max and n_small - small_done[0] >= wave_min // 2):
                            wave, ready[:] = list(ready), []
                            waves_running[0] += 1
                    if wave:
                        try:
                            order_s = [nm for nm in names if nm in set(wave)]
                            txts = mbgraph_native.sparse_flow_native(ctx_b, [small_recs[nm].graph for nm in order_s], ["%s_%s" % (sample, nm) for nm in order_s],
                                                                     seed, raw=True, threaded=True)
                        finally:
                            with small_lock:
                                waves_running[0] -= 1
                        for nm, txt in zip(order_s, txts):
                            small_recs[nm].fasta_raw = txt
                        if pstream is not None:               # (their pieces on whatever threads are free)
                            pool_ = _graph_pool(graph_threads)
                            with small_lock:
                                post_futs.extend(pool_.submit(post_piece, 1 + part_index[nm], txt) for nm, txt in zip(order_s, txts))
                tt["sparse flow"] = time.time() - t1
            if timeline is not None:
                timeline[name] = (tt.pop("_t0") - t_graph, time.time() - t_graph, dict(tt), len(part["routes"][name]))
            else:
                tt.pop("_t0", None)
            return rec, tt

        def _one_partition(name):
            tt = {"_t0": time.time()}
            t0 = time.time()
            n_kmers = unitigs.n_kmers(part_index[name]) if unitigs is not None else part["n_kmer_nodes"][name]
            cutoff = 10 * n_kmers + 1                                    # multibridging.py:26-30, 385-391
            rts = part["routes"][name]
            rdev_ = part.get("routes_dev")
            by_place = (native_graph and rows_mode and rdev_ is not None and name in rdev_[1] and isinstance(rts, kfc.RouteView) and not check_rows)
            idx = min(len(rts), cutoff) if by_place else rts[:cutoff]       # (by place: the count is all the host needs)
            if by_place and idx == 0:
                by_place, idx = False, np.zeros(0, np.uint32)
            if native_graph and rows_mode and (idx if by_place else len(idx)):
                # the reads named by their rows: distinct reads found on the device, their text decoded from the host matrices
                rb = None
                if check_rows:
                    rb_ = part["k1mer_bytes"][name]
                    rb = rb_() if callable(rb_) else rb_
                rdev = part.get("routes_dev")                             # (the same list, still on the device: idx is a prefix of the partition's)
                def run_rows(rb):
                    return mbgraph_native.run_partition_rows(ctx_b, unitigs, part_index[name], d1, d2, store.r1, store.r2 if paired else None,
                                                             idx if by_place else np.asarray(idx, dtype=np.uint32), rb if (rb is not None and len(rb)) else None,
                                                             0 if rb is None else len(rb) // (K + 1),
                                                             routes=(rdev[0], rdev[1][name]) if (rdev is not None and name in rdev[1]) else None)
                try:
                    gh = run_rows(rb)
                except _lib.ShannonError as ex:
                    if rb is not None or "needs the k1-mer rows" not in str(ex):
                        raise
                    rb_ = part["k1mer_bytes"][name]
                    gh = run_rows(rb_() if callable(rb_) else rb_)
                tt["graph"] = time.time() - t0
                return PartitionRecord(len(part["routes"][name]), part["n_k1mer_rows"][name], gh), tt
            if native_graph:
                if ss:
                    b1, o1, rc1, enc = store.gather_codes_ss(idx, 1)
                    b2, o2, rc2, _e = store.gather_codes_ss(idx, 2) if paired else (None, None, None, enc)
                    assert _e == enc, "both mates' reads must be stored the same way (strings, code matrices or codes + offsets)"
                else:
                    b1, o1, rc1, enc = store.gather_codes(idx, 1)
                if ss:
                    pass
                elif paired and rc1 is not None:
                    # the second mates are the same stored rows read on the other strand (shannon.py:413-424)
                    b2, o2, rc2 = b1, o1, (1 - rc1).astype(np.uint8)
                else:
                    b2, o2, rc2, _e = store.gather_codes(idx, 2) if paired else (None, None, None, enc)
                tt["materialize reads"] = time.time() - t0
                t0 = time.time()
                def rows_now():
                    rb_ = part["k1mer_bytes"][name]
                    return rb_() if callable(rb_) else rb_
                # with GPU unitigs the k1-mer rows are only needed for a partition holding a cycle of condensable edges (built by
                # the sequential code) and for the development check SHN_GRAPH_CHECK=1
                rb = rows_now() if (unitigs is None or check_rows) else None
                res_src = (d1, d2, np.asarray(idx, dtype=np.uint32)) if (enc == 1 and len(idx) and not ss and not getattr(store, "ragged", False)) else None     # code matrices resident on the device
                try:
                    gh = mbgraph_native.run_partition_handle(None if rb is None else (rb if len(rb) else np.zeros(1, np.uint8)),
                                                             0 if rb is None else len(rb) // (K + 1), K, b1, o1, b2, o2, ctx=ctx_b,
                                                             enc=enc, rc1=rc1, rc2=rc2, unitigs=unitigs, part=part_index[name], resident=res_src)
                except _lib.ShannonError as ex:
                    if rb is not None or "needs the k1-mer rows" not in str(ex):
                        raise
                    rb = rows_now()
                    gh = mbgraph_native.run_partition_handle(rb if len(rb) else np.zeros(1, np.uint8), len(rb) // (K + 1), K, b1, o1, b2,
                                                             o2, ctx=ctx_b, enc=enc, rc1=rc1, rc2=rc2, unitigs=unitigs, part=part_index[name],
                                                             resident=res_src)
                n_rows = part["n_k1mer_rows"][name]
                if enc == 1 and not ss:
                    store.release(b1)
                    if b2 is not None and b2 is not b1:
                        store.release(b2)
                tt["graph"] = time.time() - t0
                return PartitionRecord(len(part["routes"][name]), n_rows, gh), tt
            else:
                rows = part["k1mers"][name]
                if ss:
                    r1 = [store._get(store.r1, int(d)) for d in idx]
                    reads = [r1, [store._rc(store._get(store.r2, int(d))) for d in idx]] if paired else [r1]
                else:
                    r1 = [store.mate1(int(d)) for d in idx]
                    reads = [r1, [store.mate2(int(d)) for d in idx]] if paired else [r1]
                tt["materialize reads"] = time.time() - t0
                t0 = time.time()
                g, singles, comps = mbgraph.run_partition(rows, reads, K, paired, hits_factory)
                glog, n_rows = g.log, len(rows)
            tt["graph"] = time.time() - t0
            return {"n_reads_routed": len(part["routes"][name]), "n_k1mers": n_rows, "singles": singles, "components": comps,
                    "log": glog}, tt

        t_graph = time.time()
        results, futs = [], {}
        try:
            if native_graph and len(names) > 1 and graph_threads > 1:
                # partitions are independent (one multibridging process each in the reference, run_MB_SF_fn.py:219-253): the
                # native stage releases the GIL, its GPU sections take turns
                pool = _graph_pool(graph_threads)
                # the partitions with the most routed reads first (the reference's size-sorted job list, shannon.py:546-551)
                by_size = sorted(names, key=lambda nm: -len(part["routes"][nm]))
                futs = {nm: pool.submit(one_partition, nm) for nm in by_size}
                if pstream is not None:
                    post_futs.append(pool.submit(post_piece, 0, single_text))
                results = [futs[nm].result() for nm in names]
                for f_ in list(post_futs):
                    f_.result()
            else:
                for nm in names:
                    results.append(one_partition(nm))
        except BaseException:
            # a partition failed: nothing of this step may still run on the pool's threads when the inputs go away below
            if futs:
                import concurrent.futures as _cf
                for f_ in list(futs.values()) + list(post_futs):
                    f_.cancel()
                _cf.wait(list(futs.values()) + list(post_futs))
            if pstream is not None:
                pstream.close()
            # ... and the graphs already built go back now (device + host memory), not when the collector finds them
            built = [f.result()[0] for f in futs.values() if f.done() and not f.cancelled() and f.exception() is None] or [r[0] for r in results]
            for rec in built:
                g = getattr(rec, "graph", None)
                if g is not None:
                    g.close()
            raise
        finally:
            if unitigs is not None:
                unitigs.close()
        wall = time.time() - t_graph
        if timeline is not None:
            import sys
            for nm, (a, b, tt_, nr) in sorted(timeline.items(), key=lambda kv: -kv[1][1])[:12]:
                sys.stderr.write("[parts] %-16s start %6.2f end %6.2f  routed %8d  %s\n" % (nm, a, b, nr, {k: round(v, 2) for k, v in tt_.items()}))
            sys.stderr.write("[parts] stage wall %.2f s, %d partitions, %d threads, %.2f thread-seconds\n"
                             % (wall, len(names), graph_threads, sum(b - a for a, b, _t, _n in timeline.values())))
        busy = sum(sum(tt.values()) for _, tt in results) or 1.0
        for name, (rec, tt) in zip(names, results):
            for k_, v in tt.items():                                      # wall time of the stage, split like the thread time
                T[k_] = T.get(k_, 0.0) + v * wall / busy
            R.partitions[name] = rec
        t0 = time.time()
        if native_graph and os.environ.get("SHN_SFLOW_NATIVE", "1") != "0":
            # all components of all partitions through the native sparse-flow stage (shn_sparse_flow) in one call; the graphs
            # stay native objects (exported to Python tables only if somebody asks a PartitionRecord for them)
            if sflow_beside and all(getattr(R.partitions[nm], "fasta_raw", None) is not None for nm in names):
                texts = [R.partitions[nm].fasta_raw for nm in names]
            else:
                texts = mbgraph_native.sparse_flow_native(ctx_b, [R.partitions[nm].graph for nm in names], ["%s_%s" % (sample, nm) for nm in names], seed,
                                                          raw=True)
                for name, txt in zip(names, texts):
                    R.partitions[name].fasta_raw = txt                 # decoded when somebody reads ["reconstructed_fasta"]
            tick("sparse flow", t0)
            t0 = time.time()
            R._texts = [single_text] + texts                           # all_reconstructed.fasta: single contigs, then the partitions
            streamed = pstream is not None and not ps_failed and len(pstream.keep) == len(names) + 1
            if pstream is not None and not streamed:
                pstream.close()
            if os.environ.get("SHN_POST_NATIVE", "1") != "0":
                try:
                    R.final = (pstream.finish(double_stranded, lazy=True) if streamed else
                               post.finalize_texts(R._texts, double_stranded, ctx=ctx_b, lazy=True))
                except _lib.ShannonError as ex:
                    if "non-ACGT" not in str(ex) and "empty line" not in str(ex):
                        raise
                    import sys
                    sys.stderr.write("[shannon_amd] the native merge declined the transcripts (%s): merging with the Python form "
                                     "(post.finalize), about 2x slower\n" % str(ex)[:200])
                    T["post fell back to python"] = T.get("post fell back to python", 0) + 1
                    R.final = post.finalize(R.all_reconstructed, double_stranded)
            else:
                R.final = post.finalize(R.all_reconstructed, double_stranded)
            tick("post", t0)
            R.timings = T
            return R
        lines = (single_text if isinstance(single_text, str) else bytes(single_text).decode()).splitlines(True)
        for name in names:
            rec = R.partitions[name]
            sf_jobs.append((name, rec["singles"], rec["components"]))
        flat = [(nd["nodes"], nd["edges"], nd["paths"]) for _, _, comps in sf_jobs for nd in comps]
        # component c of partition p uses RNG stream id = its index within the partition (as one
        # algorithm_SF.py process per component, run_MB_SF_fn.py:242-250)
        trs_flat = []
        if flat:
            ids, gens_in = [], []
            for _, _, comps in sf_jobs:
                for c, nd in enumerate(comps):
                    ids.append(c)
            trs_flat = _sparse_flow_with_ids(ctx_b, flat, ids, seed)
        k = 0
        for name, singles, comps in sf_jobs:
            sname = "%s_%s" % (sample, name)
            txt = ""
            for c in range(len(comps)):
                txt += sparse_flow.fasta_records(sname, str(c), trs_flat[k])
                k += 1
            txt += sparse_flow.single_nodes_fasta(sname, singles)
            R.partitions[name]["reconstructed_fasta"] = txt
            lines.extend(txt.splitlines(True))
        tick("sparse flow", t0)
        t0 = time.time()
        R.all_reconstructed = lines
        R.final = post.finalize(lines, double_stranded)
        tick("post", t0)
        R.timings = T
        return R

    if defer_back == "early":
        # the step cut behind the extension: count + extension now, everything else in the returned function (two batches in flight:
        # the halves are then 1.0 s and 0.5 s of a configs[2] step instead of 1.15 s and 0.35 s)
        def rest(bctx=None):
            middle(bctx if bctx is not None else ctx)
            return back(bctx)
        return rest
    middle(ctx)
    if defer_back:
        return back
    return back()


def _sparse_flow_with_ids(ctx, components, comp_ids, seed):
    gens = [sparse_flow.component_coroutine(nd, ed, pt, cid) for (nd, ed, pt), cid in zip(components, comp_ids)]
    results = [None] * len(gens)
    pending = {}
    for c, g in enumerate(gens):
        try:
            pending[c] = next(g)
        except StopIteration as stop:
            results[c] = stop.value
    while pending:
        order = sorted(pending)
        xs = sparse_flow.solve_batch(ctx, [pending[c] for c in order], seed)
        nxt = {}
        for c, x in zip(order, xs):
            try:
                nxt[c] = gens[c].send(sparse_flow.finish(pending[c], x))
            except StopIteration as stop:
                results[c] = stop.value
        pending = nxt
    return results
