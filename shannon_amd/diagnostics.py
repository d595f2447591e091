"""Run-to-run determinism diagnostics: count -> extension -> contig stage with a checksum per stage.

staged_run() counts the reads, runs the walks with SHN_EXT_DIGEST=1 (shn_extend keeps checksums of its arrays: 8 stages x 64
chunks, shn_ext_digests) and goes through the post-walk path of run_correction step by step (live stats, accept filter, emit,
contig stage), hashing what every step returns.  first_difference() names the FIRST stage at which two such runs differ and,
for device arrays, which 64ths of the array; for host arrays the first differing index.  Used by tests/test_stress_gpu.py and
tools/stress_digest.py: the reference's loop (extension_correction.py:334-397) is sequential and deterministic, so every
repeat on the same input must agree at every stage."""
import hashlib, os
import numpy as np

STAGES = ["table.keys", "table.counts", "table.bucket_off", "ext.weights+flags", "ext.records", "ext.seed_order", "ext.claims", "ext.walk_records"]


def h(*arrs):
    m = hashlib.sha256()
    for a in arrs:
        if isinstance(a, (list, tuple)) and a and isinstance(a[0], str):
            m.update("\n".join(a).encode())
        elif isinstance(a, str):
            m.update(a.encode())
        else:
            m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()[:16]


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return "shapes %s / %s" % (a.shape, b.shape)
    d = np.nonzero(a != b)[0]
    return "equal" if not len(d) else "first differing index %d of %d (%d differ): %s / %s" % (d[0], len(a), len(d), a[d[0]], b[d[0]])


def staged_run(ctx, sets, K):
    """[(stage name, digest, payload kept for the diff)] in pipeline order"""
    from . import device, extension_correction as ec, _lib
    out = []
    old = os.environ.get("SHN_EXT_DIGEST")
    os.environ["SHN_EXT_DIGEST"] = "1"
    try:
        return _staged_run(ctx, sets, K, device, ec, _lib, out)
    finally:
        if old is None:
            os.environ.pop("SHN_EXT_DIGEST", None)
        else:
            os.environ["SHN_EXT_DIGEST"] = old


def _staged_run(ctx, sets, K, device, ec, _lib, out):
    t = device.count_k1mers(ctx, sets, K + 1)
    keys, cnts = t.download()
    order = np.argsort(keys, kind="stable")
    out.append(("count.multiset", h(keys[order], cnts[order]), (keys[order], cnts[order])))
    out.append(("count.layout", h(keys, cnts), (keys, cnts)))
    ext = ec.Extension(ctx, t, 3)
    dig = np.zeros(512, np.uint64)
    _lib.check(_lib.lib().shn_ext_digests(ext.h, dig.ctypes.data))
    dig = dig.reshape(8, 64)
    for i, name in enumerate(STAGES):
        out.append((name, h(dig[i]), dig[i].copy()))
    k1 = K + 1
    live, nr, nl, tw = ext.live_stats(75 - k1)
    out.append(("live_stats", h(live, nr, nl, tw), (live.copy(), nr.copy(), nl.copy(), tw.copy())))
    keep = ec.accept_filter(live, nr, nl, tw, k1, 75, 3)
    kr = np.array([x[0] for x in keep], np.int64)
    kl = np.array([x[1] for x in keep], np.int64)
    out.append(("accept_filter", h(kr, kl), (kr, kl)))
    strings = ext.emit([x[0] for x in keep], [x[1] for x in keep]) if keep else []
    out.append(("emit", h(strings), strings))
    acc, coff, cnb, cw = ec.contig_stage(strings, k1)
    out.append(("contig_stage", h(acc, np.asarray(coff, np.int64), np.asarray(cnb, np.int64), np.asarray(cw, np.int64)),
                (np.asarray(acc), np.asarray(coff, np.int64), np.asarray(cnb, np.int64), np.asarray(cw, np.int64))))
    info = dict(n_table=len(keys), n_walks=ext.n_walks, rounds=ext.iterations, candidates=len(strings), accepted=int(np.count_nonzero(acc)))
    ext.close()
    t.close()
    return out, info


def describe(name, a, b):
    if name.startswith("table.") or name.startswith("ext."):
        d = np.nonzero(a != b)[0]
        return "chunks that differ (of 64): %s" % d.tolist()
    if isinstance(a, list):
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                return "string %d of %d / %d differs (lengths %d / %d)" % (i, len(a), len(b), len(x), len(y))
        return "list lengths %d / %d" % (len(a), len(b))
    return "; ".join(first_diff(x, y) for x, y in zip(a, b))




def first_difference(first, run):
    """None if the two staged runs agree, else a one-line description: the first stage that differs and where"""
    for (name, dg, pay), (_n, dg0, pay0) in zip(run, first):
        if dg != dg0:
            later = [n for (n, d, _p), (_n2, d0, _p0) in zip(run, first) if d != d0]
            return "first differing stage: %s [%s]; all differing stages: %s" % (name, describe(name, pay0, pay), later)
    return None
