"""Compiles the oracle's C restatements (gcc) into oracle/_build/.  Test infrastructure."""
import os, subprocess, ctypes
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboracle.so")


def build(force=False):
    src = os.path.join(HERE, "count_c.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        os.makedirs(OUT, exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", LIB, src], check=True)
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
