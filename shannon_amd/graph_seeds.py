"""HIP K-mer seed scans for the graph stage (csrc/seeds.hip) plugged into shannon_amd/mbgraph.py
through its `seed_hits` hooks: the distinct reads of the partition are packed on the device once,
the per-read / per-offset probing of find_bridging_reads (mbgraph.py:88-111) and known_paths
(mbgraph.py:1355-1388) runs as kernels, the few hits come back to the host logic."""
import ctypes as C
import numpy as np
from . import _lib, device
from .extension_correction import windows_to_keys
from .kmers_for_component import make_table


_CODE = np.full(256, 0, dtype=np.uint64)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def pack_strings(keys, K):
    """Packed uint64 keys of equal-length ACGT strings (vectorised)."""
    n = len(keys)
    arr = _CODE[np.frombuffer("".join(keys).encode(), dtype=np.uint8).reshape(n, K)]
    out = np.zeros(n, dtype=np.uint64)
    for j in range(K):
        out = (out << np.uint64(2)) | arr[:, j]
    return out


def hits_factory(ctx):
    """Returns f(graph) -> (bridging_hits, path_hits) for mbgraph.run_partition."""

    def factory(g):
        state = {"reads": None}

        def dev_reads():
            if state["reads"] is None:
                state["reads"] = device.Reads.from_strings(ctx, g.rbases)
            return state["reads"]

        def bridging(starts):
            if not starts or not g.rbases:
                return
            keys = list(starts)
            tab = make_table(ctx, pack_strings(keys, g.K), np.arange(1, len(keys) + 1, dtype=np.uint32), g.K)
            rd = dev_reads()
            n = C.c_uint64(0)
            _lib.check(_lib.lib().shn_seed_scan(ctx.h, rd.h, g.K, tab.h, C.byref(n), None, None, None))
            if n.value:
                r = np.empty(n.value, np.uint32)
                s = np.empty(n.value, np.uint32)
                i = np.empty(n.value, np.uint32)
                _lib.check(_lib.lib().shn_seed_scan(ctx.h, rd.h, g.K, tab.h, C.byref(n), r.ctypes.data, s.ctypes.data, i.ctypes.data))
                for rr, ss, ii in zip(r.tolist(), s.tolist(), i.tolist()):
                    yield rr, ss, starts[keys[ii]]
            tab.close()

        def paths(kmers):
            if not kmers or not g.rbases:
                return
            keys = list(kmers)
            tab = make_table(ctx, pack_strings(keys, g.K), np.arange(1, len(keys) + 1, dtype=np.uint32), g.K)
            rd = dev_reads()
            a = np.zeros(len(g.rbases), np.uint32)
            b = np.zeros(len(g.rbases), np.uint32)
            _lib.check(_lib.lib().shn_seed_ends(ctx.h, rd.h, g.K, tab.h, a.ctypes.data, b.ctypes.data))
            tab.close()
            rd.close()
            state["reads"] = None
            for r in np.nonzero((a > 0) & (b > 0))[0].tolist():
                yield r, kmers[keys[a[r] - 1]]

        return bridging, paths

    return factory
