"""Builds libshannon_hip.so (hipcc, gfx950 only) in-tree.  `python -m shannon_amd.build`."""
import os, subprocess, sys, glob

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libshannon_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(PKG, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unused-value", "-Wno-unused-result", "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
