"""shannon_amd -- MI355X-native (gfx950) implementation of the Shannon RNA-Seq assembler hot
path: (K+1)-mer counting -> contig / multibridged de-Bruijn graph -> sparse-flow path
decomposition.  Python host layer over a C-ABI shared library of hand-written HIP kernels
(shannon_amd/csrc, include/shannon_hip.h)."""
__version__ = "0.1.0"
