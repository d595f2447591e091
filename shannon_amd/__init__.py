"""shannon_amd -- MI355X-native (gfx950) implementation of the Shannon RNA-Seq assembler hot
path: (K+1)-mer counting -> contig / multibridged de-Bruijn graph -> sparse-flow path
decomposition.  Python host layer over a C-ABI shared library of hand-written HIP kernels
(shannon_amd/csrc, include/shannon_hip.h)."""
__version__ = "0.1.0"


def malloc_tune():
    """OPT-IN process-wide allocator settings for programs that own their process (bench.py, shannon.py call this; an application
    that embeds the package or the library is left alone): blocks up to 32 MB come from the heap and freed memory stays there, so
    that the host stages do not pay for fresh zero pages every step (~4 % of a step at BASELINE configs[2]).  Also applied at
    import / library load when SHN_MALLOC_TUNE=1 is in the environment; MALLOC_*_ environment settings win."""
    import ctypes, os
    try:
        libc = ctypes.CDLL(None)
        M_TRIM_THRESHOLD, M_TOP_PAD, M_MMAP_THRESHOLD = -1, -2, -3
        if "MALLOC_MMAP_THRESHOLD_" not in os.environ:
            libc.mallopt(M_MMAP_THRESHOLD, 32 << 20)
        if "MALLOC_TRIM_THRESHOLD_" not in os.environ:
            libc.mallopt(M_TRIM_THRESHOLD, -1)                    # (size_t) -1: the heap is never trimmed
        if "MALLOC_TOP_PAD_" not in os.environ:
            libc.mallopt(M_TOP_PAD, 256 << 20)
    except (OSError, AttributeError):
        pass


import os as _os
if _os.environ.get("SHN_MALLOC_TUNE", "0") == "1":
    malloc_tune()
