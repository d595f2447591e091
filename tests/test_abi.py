"""CPU: the C-ABI library builds, loads, and exports every symbol include/shannon_hip.h declares."""
import ctypes, os, re
from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "shannon_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(shn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from shannon_amd import build, _lib
    build.build(verbose=False)
    l = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(l, s), "missing export " + s
    assert sorted(_lib.SIGNATURES) == syms
    l.shn_version.restype = ctypes.c_char_p
    assert b"gfx950" in l.shn_version()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from shannon_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    import pytest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
