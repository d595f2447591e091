"""Compiles the oracle's C restatements (gcc) into oracle/_build/.  Test infrastructure."""
import os, subprocess, ctypes
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboracle.so")


SOURCES = ["count_c.c", "ext_c.c"]


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in SOURCES]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        os.makedirs(OUT, exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", LIB] + srcs, check=True)
    return LIB


def count_canonical(codes, k1, canonical=True):
    """codes: uint8 [n, L].  Returns (sorted canonical keys, counts, n_windows) via the C oracle."""
    lib = ctypes.CDLL(build())
    f = lib.oracle_count_canonical
    f.restype = ctypes.c_uint64
    f.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, L = codes.shape
    cap = max(1, n * max(0, L - k1 + 1))
    keys = np.empty(cap, dtype=np.uint64)
    cnts = np.empty(cap, dtype=np.uint32)
    nw = ctypes.c_uint64(0)
    d = f(codes.ctypes.data, n, L, k1, 1 if canonical else 0, keys.ctypes.data, cnts.ctypes.data, ctypes.byref(nw))
    return keys[:d].copy(), cnts[:d].copy(), nw.value


def extend(keys, counts, k1, min_weight=3, strings=True):
    """The greedy extension (rows a3-a4) over a canonical count table through the C oracle (ext_c.c).
    keys: canonical keys ascending (uint64), counts (uint32).  Returns the walks in seed order as a list of
    (contig string, total weight, number of k1-mers) -- what extension_correction.py:343-354 derives per seed.
    strings=False: the raw arrays (seed keys, right steps, left steps, total weights, step bases) instead."""
    lib = ctypes.CDLL(build())
    f = lib.oracle_extend
    f.restype = ctypes.c_uint64
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    n = len(keys)
    cap = 2 * n + 2
    seed = np.empty(cap, np.uint64); nr = np.empty(cap, np.uint32); nl = np.empty(cap, np.uint32); tw = np.empty(cap, np.uint64)
    bases = np.empty(cap, np.uint8)
    ne = ctypes.c_uint64(0)
    nw = int(f(keys.ctypes.data, counts.ctypes.data, n, int(k1), int(min_weight), seed.ctypes.data, nr.ctypes.data, nl.ctypes.data,
               tw.ctypes.data, bases.ctypes.data, ctypes.byref(ne)))
    if not strings:
        return seed[:nw], nr[:nw], nl[:nw], tw[:nw], bases[:int(nr[:nw].sum(dtype=np.uint64) + nl[:nw].sum(dtype=np.uint64))]
    A = np.frombuffer(b"ACGT", np.uint8)
    out, at = [], 0
    for i in range(nw):
        a, b = int(nr[i]), int(nl[i])
        s = int(seed[i])
        start = "".join("ACGT"[(s >> (2 * (k1 - 1 - j))) & 3] for j in range(k1))
        right = A[bases[at:at + a]].tobytes().decode()
        left = A[bases[at + a:at + a + b][::-1]].tobytes().decode()
        at += a + b
        out.append((left + start + right, int(tw[i]), a + b + 1))
    return out
