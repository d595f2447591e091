"""Host-side helpers of the C ABI that need no GPU: threaded row gather, k-window enumeration."""
import numpy as np
import pytest


def test_gather_rows_matches_numpy_and_checks_its_indices():
    from shannon_amd import _lib, build
    build.build(verbose=False)
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (5000, 100), dtype=np.uint8)
    for n in (0, 1, 7, 40000):
        idx = rng.integers(0, 5000, n)
        assert np.array_equal(_lib.gather_rows(src, idx, threads=8), src[idx])
    with pytest.raises(_lib.ShannonError):
        _lib.gather_rows(src, np.array([5000]))
    with pytest.raises(_lib.ShannonError):
        _lib.gather_rows(src, np.array([-1]))


def test_string_windows_match_the_vectorised_path():
    from shannon_amd import _lib, build
    from shannon_amd.extension_correction import windows_to_keys, windows_to_keys_many
    build.build(verbose=False)
    rng = np.random.default_rng(1)
    strings = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(n))) for n in rng.integers(0, 400, 300)] + ["", "ACG"]
    for k in (1, 15, 26, 32):
        keys, rows, nwin = _lib.string_windows(strings, k, want_keys=True, want_rows=True)
        exp = [windows_to_keys(c, k) for c in strings if len(c) >= k]
        exp = np.concatenate(exp) if exp else np.zeros(0, np.uint64)
        assert np.array_equal(keys, exp) and int(nwin.sum()) == len(exp)
        exp_rows = [np.lib.stride_tricks.sliding_window_view(np.frombuffer(c.encode(), np.uint8), k).reshape(-1) for c in strings if len(c) >= k]
        assert np.array_equal(rows, np.concatenate(exp_rows) if exp_rows else np.zeros(0, np.uint8))
        k2, nw2 = windows_to_keys_many(strings, k)                       # (native above 4096 bases, numpy below)
        assert np.array_equal(k2, exp) and np.array_equal(nw2, nwin)
    with pytest.raises(_lib.ShannonError):
        _lib.string_windows(["ACGTNACGT"], 3)
