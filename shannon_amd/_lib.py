"""ctypes binding of libshannon_hip.so (include/shannon_hip.h).  The product path has NO CPU
fallback: if the library is missing or a call fails, a RuntimeError is raised."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHN_HIP_LIB") or os.path.join(_PKG, "libshannon_hip.so")     # SHN_HIP_LIB: A/B builds of the same ABI
_lib = None

u8p, u32p, u64p, dblp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double)
vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)

SIGNATURES = {
    "shn_last_error": (C.c_char_p, []),
    "shn_version": (C.c_char_p, []),
    "shn_ctx_create": (C.c_int, [C.c_int, vp, vpp]),
    "shn_ctx_destroy": (None, [vp]),
    "shn_ctx_sync": (C.c_int, [vp]),
    "shn_ctx_fork": (C.c_int, [vp, vpp]),
    "shn_ctx_own_workspaces": (C.c_int, [vp]),
    "shn_timer_reset": (C.c_int, [vp]),
    "shn_timer_ms": (C.c_int, [vp, C.c_int, dblp, u64p]),
    "shn_timer_bytes": (C.c_int, [vp, C.c_int, u64p]),
    "shn_timer_name": (C.c_char_p, [C.c_int]),
    "shn_reads_create": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vpp]),
    "shn_reads_destroy": (None, [vp]),
    "shn_reads_count": (C.c_uint64, [vp]),
    "shn_reads_total_bases": (C.c_uint64, [vp]),
    "shn_reads_max_len": (C.c_uint32, [vp]),
    "shn_reads_n_invalid": (C.c_uint64, [vp]),
    "shn_string_windows": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, vp]),
    "shn_gather_rows": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, C.c_int]),
    "shn_gather_segments": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.c_int]),
    "shn_fasta_records": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_char_p, vp, C.c_uint64, u64p]),
    "shn_count_k1mers": (C.c_int, [vp, vpp, C.c_int, C.c_int, C.c_int, vpp]),
    "shn_table_destroy": (None, [vp]),
    "shn_table_size": (C.c_uint64, [vp]),
    "shn_table_total": (C.c_uint64, [vp]),
    "shn_table_k": (C.c_int, [vp]),
    "shn_table_canonical": (C.c_int, [vp]),
    "shn_table_download": (C.c_int, [vp, vp, vp, vp]),
    "shn_table_dump": (C.c_int, [vp, vp, C.c_uint32, vp, vp, u64p]),
    "shn_table_filter_lower": (C.c_int, [vp, vp, C.c_uint32, vpp]),
    "shn_table_device_ptrs": (C.c_int, [vp, vpp, vpp]),
    "shn_table_lookup": (C.c_int, [vp, vp, vp, C.c_uint64, vp]),
    "shn_table_from_pairs": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vpp]),
    "shn_table_merge_rc": (C.c_int, [vp, vp, vp, vpp]),
    "shn_table_shard": (C.c_int, [vp, vp, C.c_int, u64p, vp, vp]),
    "shn_table_shard_mode": (C.c_int, [vp, vp, C.c_int, C.c_int, u64p, vp, vp]),
    "shn_debug_cc_counters": (C.c_int, [u64p, C.c_int]),
    "shn_cc_create": (C.c_int, [vp, vp, C.c_int, C.c_int, vpp]),
    "shn_cc_destroy": (None, [vp]),
    "shn_cc_query_counts": (C.c_int, [vp, u64p]),
    "shn_cc_queries": (C.c_int, [vp, vp, vp]),
    "shn_cc_answer": (C.c_int, [vp, vp, vp, u64p, u64p, vp, u64p]),
    "shn_cc_solve": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp, vp, u64p]),
    "shn_cc_labels": (C.c_int, [vp, C.c_uint64, vp, vp, C.c_uint64, vp]),
    "shn_cc_owners": (C.c_int, [vp, vp, vp, vp, C.c_uint64, vp]),
    "shn_cc_shard": (C.c_int, [vp, vp, u64p, vp, vp]),
    "shn_table_create": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vpp]),
    "shn_probe_build": (C.c_int, [vp, vp, vp, C.c_uint64, vp, C.c_uint32, C.c_int, vpp]),
    "shn_probe_destroy": (None, [vp]),
    "shn_probe_table": (vp, [vp]),
    "shn_probe_n_sets": (C.c_uint32, [vp]),
    "shn_probe_n_members": (C.c_uint64, [vp]),
    "shn_probe_sets": (C.c_int, [vp, vp, vp]),
    "shn_route_reads": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_uint32, vpp]),
    "shn_route_reads_mode": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_uint32, C.c_int, vpp]),
    "shn_routes_destroy": (None, [vp]),
    "shn_routes_size": (C.c_uint64, [vp]),
    "shn_routes_download": (C.c_int, [vp, vp, vp, vp]),
    "shn_routes_bounds": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, vp, vp]),
    "shn_routes_download_range": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp]),
    "shn_lp_solve_batch": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "shn_lp_set_rule": (C.c_int, [vp, C.c_int]),
    "shn_lp_stats": (C.c_int, [vp, vp, C.c_int]),
    "shn_contig_graph": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_double, vp, u64p, vp, vp, vp, u64p]),
    "shn_seed_scan": (C.c_int, [vp, vp, C.c_int, vp, u64p, vp, vp, vp]),
    "shn_rmer_join": (C.c_int, [vp, vp, vp, C.c_int, u64p, vp, vp, vp]),
    "shn_contig_best_counts": (C.c_int, [vp, C.c_uint64]),
    "shn_seed_ends": (C.c_int, [vp, vp, C.c_int, vp, vp, vp]),
    "shn_mbgraph_run": (C.c_int, [vp, C.c_int, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vpp]),
    "shn_graph_destroy": (None, [vp]),
    "shn_unitigs_build": (C.c_int, [vp, vp, vp, C.c_uint64, vp, C.c_uint32, C.c_int, vpp]),
    "shn_unitigs_destroy": (None, [vp]),
    "shn_unitigs_n_kmers": (C.c_uint64, [vp, C.c_uint32]),
    "shn_mbgraph_run_unitigs": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vpp]),
    "shn_graph_sizes": (C.c_int, [vp, u64p]),
    "shn_graph_export": (C.c_int, [vp] + [vp] * 20),
    "shn_mbgraph_run_rows": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, C.c_int, vpp]),
    "shn_mbgraph_run_routes": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_int, vpp]),
    "shn_host_cpus": (C.c_int, []),
    "shn_partition_metis": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp]),
    "shn_partition_csr": (C.c_int, [C.c_uint64, vp, vp, vp, C.c_int, C.c_int, vp]),
    "shn_metis_reweight": (C.c_int, [C.c_char_p, C.c_uint64, vp, C.c_uint64, C.c_int, vp, C.c_uint64, u64p]),
    "shn_malloc_tune_now": (None, []),
    "shn_known_paths_scan": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_uint64, vp, vp, vp]),
    "shn_known_paths_search": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "shn_post_finalize": (C.c_int, [vp, C.c_uint64, C.c_int, C.c_int, vpp]),
    "shn_post_finalize_bufs": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_int, vpp]),
    "shn_post_finalize_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vpp]),
    "shn_post_stream_begin": (C.c_int, [vp, C.c_uint64, vpp]),
    "shn_post_stream_add": (C.c_int, [vp, C.c_uint64, vp, C.c_uint64]),
    "shn_post_stream_finish": (C.c_int, [vp, C.c_int, C.c_int, vpp]),
    "shn_post_stream_destroy": (None, [vp]),
    "shn_post_count": (C.c_uint64, [vp]),
    "shn_post_sizes": (C.c_int, [vp, vp, vp]),
    "shn_post_export": (C.c_int, [vp, vp, vp, vp, vp]),
    "shn_post_destroy": (None, [vp]),
    "shn_reads_ingest": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, vpp]),
    "shn_reads_ingest_ragged": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vpp]),
    "shn_text_records_in_range": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, u64p, u64p, C.c_uint64, vp, C.c_uint64, u64p]),
    "shn_text_skip_records": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, u64p]),
    "shn_reads_dedup": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_int, vp, vp, vp, vp, vp]),
    "shn_mbgraph_run_resident": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, vpp]),
    "shn_reads_gather": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vpp]),
    "shn_graph_from_tables": (C.c_int, [vp] * 20 + [vpp]),
    "shn_sparse_flow": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, vpp]),
    "shn_sparse_flow_thread": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint64, vpp]),
    "shn_sflow_destroy": (None, [vp]),
    "shn_sflow_text_size": (C.c_uint64, [vp, C.c_uint32]),
    "shn_sflow_text": (C.c_int, [vp, C.c_uint32, vp]),
    "shn_find_reps": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp]),
    "shn_extend": (C.c_int, [vp, vp, C.c_uint32, C.c_int, vpp]),
    "shn_ext_destroy": (None, [vp]),
    "shn_ext_n_walks": (C.c_uint64, [vp]),
    "shn_ext_iterations": (C.c_int, [vp]),
    "shn_ext_total_steps": (C.c_uint64, [vp]),
    "shn_ext_wave_steps": (C.c_uint64, [vp]),
    "shn_ext_dense_rounds": (C.c_int, [vp]),
    "shn_ext_settled_walks": (C.c_uint64, [vp]),
    "shn_ext_fresh_steps": (C.c_uint64, [vp]),
    "shn_ext_digests": (C.c_int, [vp, vp]),
    "shn_debug_counter": (C.c_uint64, [C.c_int]),
    "shn_debug_alloc": (C.c_int, [vp, C.c_uint64, vpp]),
    "shn_debug_free": (None, [vp, vp]),
    "shn_debug_fill": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32]),
    "shn_debug_read": (C.c_int, [vp, vp, C.c_uint64, vp]),
    "shn_extend_sharded": (C.c_int, [vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, vpp]),
    "shn_ext_seed_info": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp]),
    "shn_ext_live_stats": (C.c_int, [vp, vp, u64p, vp, vp, vp, vp]),
    "shn_ext_live_stats_min": (C.c_int, [vp, vp, C.c_uint32, u64p, vp, vp, vp, vp]),
    "shn_ext_accept": (C.c_int, [vp, vp, C.c_uint32, C.c_double, u64p, vp, vp, vp, vp]),
    "shn_ext_stats": (C.c_int, [vp, vp, vp, vp, vp]),
    "shn_ext_stats_range": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp]),
    "shn_ext_set_block_callback": (None, [vp, vp]),
    "shn_cgraph_create": (C.c_int, [C.c_int, C.c_int, C.c_double, vpp]),
    "shn_cgraph_destroy": (None, [vp]),
    "shn_cgraph_add": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp]),
    "shn_cgraph_sizes": (C.c_int, [vp, vp, vp]),
    "shn_cgraph_export": (C.c_int, [vp, vp, vp, vp]),
    "shn_contig_stage": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_double, vp, vp, vpp]),
    "shn_contig_components": (C.c_int, [C.c_uint64, vp, vp, vp, vp, vp, vp, u64p]),
    "shn_ext_emit": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp]),
    "shn_ext_emit_device": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vpp]),
    "shn_contig_stage_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_double, vp, vp, vpp]),
    "shn_devtext_segments": (C.c_int, [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp]),
    "shn_devtext_destroy": (None, [vp]),
    "shn_ext_weights": (C.c_int, [vp, vp, vp, C.c_uint64, vp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libshannon_hip.so not found at %s -- build it with `python -m shannon_amd.build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # torch bundles its own HIP runtime (soname libamdhip64.so.7); it must be the one already
        # loaded when our library resolves libamdhip64.so.7, or two runtimes fight over the GPU.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


class ShannonError(RuntimeError):
    """a call into libshannon_hip returned an error code"""


def check(rc):
    if rc != 0:
        raise ShannonError("libshannon_hip: %s (code %d)" % (lib().shn_last_error().decode(), rc))


def host_cpus():
    """CPUs this process may keep busy (hardware threads, affinity mask, cgroup quota; SHN_HOST_CPUS overrides)."""
    return int(lib().shn_host_cpus())


def gather_rows(src, idx, out=None, threads=8):
    """out[i] = src[idx[i]] for a C-contiguous 2-D uint8 matrix, on host threads (shn_gather_rows)."""
    import numpy as np
    src = np.ascontiguousarray(src)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if out is None:
        out = np.empty((len(idx), src.shape[1]), dtype=src.dtype)
    assert src.ndim == 2 and out.flags["C_CONTIGUOUS"] and out.shape == (len(idx), src.shape[1]) and out.dtype == src.dtype
    rb = src.shape[1] * src.itemsize
    if len(idx) and rb:
        check(lib().shn_gather_rows(src.ctypes.data, src.shape[0], rb, idx.ctypes.data, len(idx), out.ctypes.data, int(threads)))
    return out


def gather_segments(src, src_off, order, threads=8):
    """(dst, dst_off): segment i of dst = segment order[i] of src (byte segments given by offsets), on host threads
    (shn_gather_segments)."""
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.uint8)
    src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
    order = np.ascontiguousarray(order, dtype=np.int64)
    lens = (src_off[1:] - src_off[:-1])[order] if len(order) else np.zeros(0, np.uint64)
    dst_off = np.zeros(len(order) + 1, dtype=np.uint64)
    dst_off[1:] = np.cumsum(lens, dtype=np.uint64)
    dst = np.empty(int(dst_off[-1]), dtype=np.uint8)
    if len(order):
        check(lib().shn_gather_segments(src.ctypes.data, src_off.ctypes.data, len(src_off) - 1, order.ctypes.data, len(order), dst.ctypes.data,
                                        dst_off.ctypes.data, int(threads)))
    return dst, dst_off


def fasta_records(src, src_off, order, prefix):
    """uint8 array: the two-line FASTA whose record i is ">" + prefix + str(i), then segment order[i] of src (shn_fasta_records)"""
    import numpy as np
    src = np.ascontiguousarray(src, dtype=np.uint8)
    src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
    order = np.ascontiguousarray(order, dtype=np.int64)
    n = C.c_uint64(0)
    args = (src.ctypes.data, src_off.ctypes.data, len(src_off) - 1, order.ctypes.data, len(order), prefix.encode())
    check(lib().shn_fasta_records(*args, None, 0, C.byref(n)))
    dst = np.empty(max(int(n.value), 1), dtype=np.uint8)
    check(lib().shn_fasta_records(*args, dst.ctypes.data, int(n.value), C.byref(n)))
    return dst[:int(n.value)]


def string_windows(strings, k, want_keys=True, want_rows=False):
    """(keys uint64[n_windows] or None, rows uint8[n_windows * k] or None, windows per string) of all k-windows of ACGT
    strings, in order (shn_string_windows).  Raises ShannonError on a base outside ACGT."""
    import numpy as np
    lens = np.array([len(c) for c in strings], dtype=np.int64)
    nwin = np.maximum(lens - k + 1, 0)
    total = int(nwin.sum())
    text = np.frombuffer("".join(strings).encode(), dtype=np.uint8)
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    keys = np.empty(total, dtype=np.uint64) if want_keys else None
    rows = np.empty(total * k, dtype=np.uint8) if want_rows else None
    if total:
        check(lib().shn_string_windows(text.ctypes.data, off.ctypes.data, len(strings), int(k),
                                       keys.ctypes.data if want_keys else None, rows.ctypes.data if want_rows else None))
    return keys, rows, nwin
