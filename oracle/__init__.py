"""CPU oracle for the Shannon hot path -- TEST INFRASTRUCTURE ONLY.

A plain-Python / numpy / C restatement of the reference algorithms (sreeramkannan/Shannon)
for the path  (K+1)-mer counting -> contig extension / partition -> multibridged de-Bruijn
graph -> sparse-flow path decomposition.  Every function cites the reference file:line it
follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import anything from this package; the product (`shannon_amd/`) never does.

Parity status (see DESIGN.md "Oracle"):
  * stages a1-a27, a29-a31 : pinned against golden vectors produced by running the reference
    itself (mechanically translated to Python 3 at run time, tests/golden/make_golden.py).
  * a28 LP optimiser (cvxopt, not vendored, version unpinned, absent here): PARITY UNPINNED.
    The wrapper logic around the LP is pinned against the reference; the LP solve is pinned
    only on unique-optimum cases against scipy-HiGHS.
  * gpmetis (external METIS): PARITY UNPINNED (partition given as input in the fixtures).
"""
