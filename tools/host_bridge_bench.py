"""The native graph stage on the host alone (shn_mbgraph_run with ctx = NULL) on a few gene families at very high coverage: thousands of
X-nodes with read lists of 10^4-10^5 entries, i.e. what bridge_all of the largest partition of BASELINE configs[2] looks like, without
a GPU.  SHN_GRAPH_LAPS=0 prints the phase times (bridge_all: refresh / in / out / distribute / condense).  Used to tune the host code
of bridge_all in round 4 (4.25 -> 3.1 s on 150,000 pairs); tests/test_host_graph.py + tests/test_oracle_golden.py keep it honest.
python tools/host_bridge_bench.py [pairs]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shannon_amd import synth, mbgraph_native
from oracle import build_c
K = 25; k1 = K + 1
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
(q1, q2), iso = synth.make_dataset(n_pairs, 6, seed=5)
A = np.frombuffer(b"ACGT", np.uint8)
codes = np.concatenate([q1, q2])
keys, cnts, nw = build_c.count_canonical(codes, k1, True)
# k1-mer rows: both orientations of every canonical k1-mer with count >= 3
def dec(key):
    return "".join("ACGT"[(int(key) >> (2 * (k1 - 1 - i))) & 3] for i in range(k1))
rc = lambda s: s[::-1].translate(str.maketrans("ACGT", "TGCA"))
rows = []
for kk, c in zip(keys.tolist(), cnts.tolist()):
    if c >= 3:
        s = dec(kk); rows.append((s, c)); r = rc(s)
        if r != s: rows.append((r, c))
# reads: strand-doubled pairs as the reference writes them (R1 + RC(R2), RC(R1) + R2)
s1 = [A[r].tobytes().decode() for r in q1]; s2 = [A[r].tobytes().decode() for r in q2]
f1 = s1 + [rc(x) for x in s2]; f2 = [rc(x) for x in s1] + s2
print("rows", len(rows), "pairs", len(f1), flush=True)
os.environ["SHN_GRAPH_LAPS"] = "0"
for rep in range(2):
    t0 = time.time()
    singles, comps, log = mbgraph_native.run_partition(rows, [f1, f2], K, True)
    print("run %.3f s; comps %d" % (time.time() - t0, len(comps)), flush=True)
