"""Builds libshannon_hip.so (hipcc, gfx950 only) in-tree.  `python -m shannon_amd.build`.

Every csrc/*.hip is compiled to an object under csrc/_obj/ (only when it or a header changed, the files in
parallel) and the objects are linked into the shared library; `force=True` recompiles everything."""
import os, subprocess, sys, glob
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libshannon_hip.so")
OBJ = os.path.join(PKG, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def headers():
    return glob.glob(os.path.join(PKG, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _obj_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers())
    todo = []
    for src in sources():
        o = _obj_of(src)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), hdr_t):
            todo.append(src)

    def compile_one(src):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", _obj_of(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    keep = {_obj_of(s) for s in sources()}
    for o in glob.glob(os.path.join(OBJ, "*.o")):              # objects of sources that no longer exist
        if o not in keep:
            os.remove(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + sorted(keep)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
