"""Sparse-flow transcript reconstruction oracle (rows a25-a27, a29, a30).  Test infrastructure
(see oracle/__init__.py).  Restates algorithm_SF.py (a script whose body runs on import,
algorithm_SF.py:858-934) as functions over in-memory component tables.

Float -> text: the reference prints floats with Python-2 str() (12 significant digits) when
mbgraph writes nodes/edges files (mbgraph.py:301-306,1345-1348) and algorithm_SF re-parses them.
The translated golden harness runs under Python 3 (repr, 17 digits), so the oracle passes
floats through unchanged; the 12-digit wire format is a documented deviation.
"""
from . import lp as _lp

PATH_SPARSITY = 10          # algorithm_SF.py:31


class SNode(object):
    __slots__ = ("string", "in_edges", "out_edges", "name", "weight", "L", "orig")

    def __init__(self, string, weight, L, name):
        """algorithm_SF.py:179-190."""
        self.string, self.weight, self.L, self.name = string, weight, L, name
        self.in_edges, self.out_edges = [], []


def _key(n):
    return int(n.name.split("_")[0])


class SFGraph(object):
    def __init__(self, nodes, edges, paths, seed=0, comp_id=0, decompose=None):
        """ParseNodeFile / ParseEdgeFile / ParseKnownPathsFile, algorithm_SF.py:91-160.
        nodes: [(id, bases, copycount, norm)], edges: [(in, out, overlap, cc, norm)], paths: [[ids]]."""
        self.seed, self.comp_id = seed, comp_id
        self.decompose = decompose or _lp.path_decompose
        self.n_decompositions = 0
        self.nodes = []
        self.by_id = {}
        for nid, bases, cc, _norm in nodes:
            n = SNode(bases, float(cc), len(bases), str(nid))      # :136 L = len(bases)
            self.by_id[str(nid)] = n
            self.nodes.append(n)
        for a, b, ov, cc, norm in edges:
            s, e = self.by_id[str(a)], self.by_id[str(b)]
            s.out_edges.append([e, int(ov), float(cc), float(norm)])
            e.in_edges.append([s, int(ov), float(cc), float(norm)])
        self.known_paths = []
        self.paths_for_node = {id(n): [] for n in self.nodes}      # :96-97
        for i, p in enumerate(paths):
            lst = []
            for h in p:
                n = self.by_id[str(h)]
                lst.append(n)
                pf = self.paths_for_node[id(n)]
                if len(pf) < PATH_SPARSITY:                         # :110
                    pf.append(i)
            self.known_paths.append(lst)
        self.start = self.end = None
        self.paths_Y = []

    def find_start_and_end(self):
        """findStartAndEnd2, algorithm_SF.py:227-247."""
        S = SNode("Start_", 0, 0, "S")
        E = SNode("_End", 0, 0, "E")
        for n in self.nodes:
            if len(n.in_edges) == 0:
                n.in_edges.append([S, 0, n.weight, 0])
                S.out_edges.append([n, 0, n.weight, 0])
                S.weight += float(n.weight)
            if len(n.out_edges) == 0:
                n.out_edges.append([E, 0, n.weight, 0])
                E.in_edges.append([n, 0, n.weight, 0])
                E.weight += float(n.weight)
        self.nodes.append(S)
        self.nodes.append(E)
        self.start, self.end = S, E

    def to_be_reduced(self):
        """search(), algorithm_SF.py:357-370 with use_Y_paths=True."""
        out = []
        for n in self.nodes:
            if n.in_edges and n.out_edges and n is not self.start and n is not self.end:
                if len(n.in_edges) > 1 or len(n.out_edges) > 1:
                    if len(n.in_edges) <= 1:
                        continue
                    out.append(n)
        return out

    def support_matrix(self, node, inedges, outedges):
        """The P construction, algorithm_SF.py:438-498.  constituent_nodes[x] is always [ancestor]
        (:382, :529), and paths_for_node has keys only for original (parsed) nodes (:96-97)."""
        m, n = len(inedges), len(outedges)
        P = [[0] * n for _ in range(m)]
        pfn = self.paths_for_node
        if id(node) not in pfn:
            return P
        cnode = self.constituent[id(node)]
        for mi, in_node in enumerate(inedges):
            if id(in_node) not in pfn:
                continue
            for ni, out_node in enumerate(outedges):
                if id(out_node) not in pfn:
                    continue
                node_paths = []
                for c in cnode:
                    node_paths = node_paths + pfn[id(c)]
                if not node_paths:
                    continue
                l_node = self.constituent[id(in_node)]
                r_node = self.constituent[id(out_node)]
                cand = (set(pfn[id(l_node[-1])]) & set(node_paths) & set(pfn[id(r_node[0])])
                        & set(pfn[id(in_node)]) & set(pfn[id(out_node)]))
                for cp in cand:
                    nl = self.known_paths[cp]
                    if any(x is cnode[0] for x in nl) and any(x is cnode[-1] for x in nl):
                        tmp1 = [next(k for k, x in enumerate(nl) if x is c) for c in cnode]
                        good = all(tmp1[k - 1] + 1 == tmp1[k] for k in range(1, len(tmp1)))
                        if good:
                            l_good = r_good = True
                            l_check = min(tmp1[0], len(l_node))
                            r_check = min(len(nl) - tmp1[-1] - 1, len(r_node))
                            for y in range(l_check):
                                if nl[tmp1[0] - 1 - y].string != l_node[-1 - y].string:
                                    l_good = False
                            for y in range(r_check):
                                if nl[tmp1[-1] + 1 + y].string != r_node[y].string:
                                    r_good = False
                            if l_good and r_good:
                                P[mi][ni] = 1
        return P

    def algorithm2(self):
        """algorithm_SF.py:373-561 (use_Y_paths=True, worry_abt_unique=0)."""
        self.constituent = {id(n): [n] for n in self.nodes}
        self._keep = list(self.nodes)          # keep ids alive
        done = False
        while not done:
            idx = 0
            while idx < len(self.nodes):       # list mutated while iterating (quirk 21)
                node = self.nodes[idx]
                idx += 1
                if node is self.start or node is self.end:
                    continue
                if len(node.in_edges) <= 1:
                    continue
                if len(node.out_edges) == 0:   # :418-422 (in_edges cannot be empty here)
                    node.out_edges.append([self.end, 0, node.weight, 0])
                    self.end.in_edges.append([node, 0, node.weight, 0])
                    self.end.weight += float(node.weight)
                inedges = [e[0] for e in node.in_edges]
                outedges = [e[0] for e in node.out_edges]
                a = [float(e[2]) for e in node.in_edges]
                b = [float(e[2]) for e in node.out_edges]
                in_attr = {id(e[0]): [e[1], e[3]] for e in node.in_edges}     # later duplicates win (:426)
                out_attr = {id(e[0]): [e[1], e[3]] for e in node.out_edges}
                P = self.support_matrix(node, inedges, outedges)
                flow, _nu = self.decompose(a, b, P, seed=self.seed,
                                           pid=(self.comp_id << 20) + self.n_decompositions,
                                           sparsity=PATH_SPARSITY)
                self.n_decompositions += 1
                m, n = len(a), len(b)
                for i in range(m):
                    for j in range(n):
                        cc = flow[i][j]
                        if cc != 0:
                            oa = out_attr[id(outedges[j])]
                            ia = in_attr[id(inedges[i])]
                            nn = SNode(node.string, cc, node.L, node.name + "_[" + str(i) + "," + str(j) + "]")
                            nn.in_edges.append([inedges[i], ia[0], cc, ia[1]])
                            inedges[i].out_edges.append([nn, ia[0], cc, ia[1]])
                            nn.out_edges.append([outedges[j], oa[0], cc, oa[1]])
                            outedges[j].in_edges.append([nn, oa[0], cc, oa[1]])
                            self.nodes.append(nn)
                            self._keep.append(nn)
                            self.constituent[id(nn)] = self.constituent[id(node)]
                # delete connections (:532-543) -- removal while iterating, as in the reference
                for e in node.in_edges:
                    t = e[0]
                    for oe in t.out_edges:
                        if oe[0] is node:
                            t.out_edges.remove(oe)
                for e in node.out_edges:
                    t = e[0]
                    for ie in t.in_edges:
                        if ie[0] is node:
                            t.in_edges.remove(ie)
                k = next((q for q, x in enumerate(self.nodes) if x is node), None)
                if k is not None:
                    del self.nodes[k]
                    # Python's list iterator keeps its index: the element after the removed one is
                    # skipped when the removed element was before the cursor.
            if not self.to_be_reduced():
                done = True
            else:
                self.nodes = [x for x in self.nodes if x is not self.start and x is not self.end]
                self.nodes.sort(key=_key)
                self.nodes.append(self.end)
                self.nodes.insert(0, self.start)

    def _read_paths(self, node, s, names, overlap, sw, sn):
        """read_paths_recursive, algorithm_SF.py:564-589."""
        cur = s + node.string[overlap:]
        cn = names + "->" + node.name.split("_")[0]
        if len(node.out_edges) == 0:
            if cur[-4:] != "_End":
                return
            cur = cur[:-4]
            avg = float(sw) / sn if sn > 0 else 0
            self.paths_Y.append([cur, avg, cn])
            return
        sw += node.weight
        sn += node.L
        for e in node.out_edges:
            self._read_paths(e[0], cur, cn, int(e[1]), sw, sn)

    def read_Y_paths(self):
        """algorithm_SF.py:592-613.  Returns [(path_str, avg_wt, node_names)] with the
        'Start_' prefix stripped and empty strings dropped."""
        self.paths_Y = []
        self._read_paths(self.start, "", "", 0, 0, 0)
        out = []
        for i, (s, w, names) in enumerate(self.paths_Y):
            ps = s[6:]
            if len(ps):
                out.append((i, ps, w, names))
        return out


def sparse_flow_component(nodes, edges, paths, seed=0, comp_id=0, decompose=None):
    """Body of algorithm_SF.py:864-934 for one component.  Returns [(i, seq, avg_wt, names)]."""
    import sys
    g = SFGraph(nodes, edges, paths, seed, comp_id, decompose)
    g.find_start_and_end()
    if len(g.nodes) <= 3:                       # :878-883
        return g.read_Y_paths()
    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(max(lim, 100000))     # :20
    try:
        g.algorithm2()
        return g.read_Y_paths()
    finally:
        sys.setrecursionlimit(lim)


def fasta_records(sname, comp, transcripts):
    """Header format of algorithm_SF.py:608-609."""
    return "".join(">Shannon_%s %s_%d\t%s\t%s\n%s\n" % (sname, comp, i, str(w), names, seq)
                   for i, seq, w, names in transcripts)


def single_nodes_fasta(sname, single_rows):
    """single_nodes_to_fasta, algorithm_SF.py:74-88: rows = [(-1, bases, cc, norm)].
    Quirk reproduced: the loop does NOT skip the header line of single_nodes.txt
    ("ID\\tBases\\tCopycount\\tNormalization", multibridging.py:286), so record 0 is the bogus
    `>..._single_0\\t Copycount:Copycount` / `Bases` and real nodes are numbered from 1 (the bogus
    record is later dropped by the >=200 bp filter, process_concatenated_fasta.py:26)."""
    rows = [("ID", "Bases", "Copycount", "Normalization")] + list(single_rows)
    return "".join(">Shannon_%s_single_%d\t Copycount:%s\n%s\n" % (sname, i, str(cc), b)
                   for i, (_h, b, cc, _n) in enumerate(rows))
