"""Multibridged de-Bruijn graph oracle (rows a12-a24).  Test infrastructure (see
oracle/__init__.py).  Restates multibridging.py (loader + pipeline order) and the live
methods of mbgraph.py with instance state instead of class globals.

Pinned orders (the reference iterates Python sets of objects / tuples, whose order is
address- or hash-dependent and therefore not reproducible even by the reference itself):
  P1  wherever the reference does list(set(reads)) (mbgraph.py:239-250, 460) the (read, index)
      pairs are ordered by (read insertion number, index);
  P2  output_components: the component's node set is scanned in Node.nodes (creation) order
      for the initial topological fringe (mbgraph.py:717), edges are written sorted by
      (InID, OutID, weight), known paths sorted by their ID tuples (multibridging.py:277-325);
  P3  Read.reads / dicts iterate in insertion order (Python 3 semantics of the translated copy).
"""


class Read(object):
    __slots__ = ("rid", "bases", "copy_count", "mate", "mate_pair", "nodes")

    def __init__(self, rid, bases, cc):
        self.rid, self.bases, self.copy_count = rid, bases, cc
        self.mate, self.mate_pair, self.nodes = None, None, None

    def bridges(self, node, index):
        """mbgraph.py:77-85."""
        if index <= 0:
            return False
        if len(self.bases) <= index + len(node.bases):
            return False
        return self.bases[index:index + len(node.bases)] == node.bases


class Edge(object):
    __slots__ = ("in_node", "out_node", "weight", "copy_count")

    def __init__(self, weight, a, b):
        """mbgraph.py:169-176."""
        self.in_node, self.out_node, self.weight = a, b, weight
        a.out_edges.append(self)
        b.in_edges.append(self)
        self.copy_count = 0.0

    def destroy(self):
        """mbgraph.py:178-182."""
        self.in_node.out_edges.remove(self)
        self.out_node.in_edges.remove(self)
        self.in_node = None
        self.out_node = None


class Node(object):
    __slots__ = ("nid", "bases", "reads", "in_edges", "out_edges", "norm", "copy_count",
                 "prevalence", "count", "destroyed", "bridged", "hash")

    def __init__(self, nid, bases):
        """mbgraph.py:328-341."""
        self.nid, self.bases = nid, bases
        self.reads, self.in_edges, self.out_edges = [], [], []
        self.norm, self.copy_count, self.prevalence, self.count = 1.0, 0.0, 0.0, 1.0
        self.destroyed = False
        self.bridged = None
        self.hash = None

    def successors(self):
        return [e.out_node for e in self.out_edges]

    def predecessors(self):
        return [e.in_node for e in self.in_edges]

    def precedes(self, other):
        return any(e.out_node is other for e in self.out_edges)

    def is_xnode(self):
        return len(self.in_edges) >= 2 and len(self.out_edges) >= 2

    def average_prevalence(self):
        return self.prevalence / self.count


class MBGraph(object):
    MATE_PAIR_LENGTH = 300          # mbgraph.py:18
    MATE_PAIR_MIN_LENGTH = 0
    MATE_PAIR_MAX_HOPS = 7          # mbgraph.py:20
    PREVALENCE_THRESHOLD = 1        # mbgraph.py:324
    HAMMING_FRACTION = 0.1          # mbgraph.py:326

    def __init__(self, K, L):
        self.K, self.L = K, L
        self.SIZE_THRESHOLD = L     # multibridging.py:203
        self.nodes = []
        self.reads = {}
        self.known_paths = set()
        self.mate_paths = set()
        self.known_edges = {}
        self.mated = False
        self._nid = 0
        self.log = []

    # ---------------------------------------------------------------- construction
    def new_node(self, bases):
        n = Node(self._nid, bases)
        self._nid += 1
        self.nodes.append(n)
        return n

    def link(self, a, b, w):
        return Edge(w, a, b)

    def load_k1mers(self, rows):
        """multibridging.py:145-172 (load_single_jellyfish): rows = [(k1mer, count)] in file order."""
        K = self.K
        idx = {}
        for bases, prev in rows:
            assert K == len(bases) - 1
            k1, k2 = bases[:-1], bases[1:]
            if k1 not in idx:
                idx[k1] = self.new_node(k1)
            if k2 not in idx:
                idx[k2] = self.new_node(k2)
            e = self.link(idx[k1], idx[k2], K - 1)
            e.copy_count = round(float(prev))
        for n in self.nodes:
            n.prevalence = sum(e.weight for e in n.out_edges)      # the quirk, :171-172

    def add_read(self, bases, cc=1.0):
        """mbgraph.py:44-62 (double=False)."""
        r = self.reads.get(bases)
        if r is not None:
            r.copy_count += cc
        else:
            r = Read(len(self.reads), bases, cc)
            self.reads[bases] = r
        return r

    def load_reads(self, reads):
        """multibridging.py:22-30: keeps reads with index <= 10*#nodes."""
        cutoff = len(self.nodes) * 10
        for i, s in enumerate(reads):
            if i <= cutoff:
                self.add_read(s.upper())
            else:
                break

    def load_mated_reads(self, reads1, reads2):
        """multibridging.py:68-97."""
        cutoff = len(self.nodes) * 10
        self.mated = True
        for i, (a, b) in enumerate(zip(reads1, reads2)):
            if i > cutoff:
                break
            r1 = self.add_read(a.upper().strip())
            r2 = self.add_read(b.upper().strip())
            r1.mate_pair, r2.mate_pair = 1, 2
            r1.mate, r2.mate = r2, r1

    # ---------------------------------------------------------------- condensing
    def remove_destroyed(self):
        self.nodes = [n for n in self.nodes if not n.destroyed]

    def destroy_node(self, n):
        """mbgraph.py:423-435 with pop=False."""
        assert not n.in_edges and not n.out_edges and not n.reads
        n.destroyed = True

    def full_destroy(self, n):
        """mbgraph.py:436-446."""
        for e in list(n.in_edges):
            e.destroy()
        for e in list(n.out_edges):
            e.destroy()
        n.reads = []
        self.destroy_node(n)

    def condense_edge(self, edge):
        """mbgraph.py:184-257 (Edge.condense, pop=False)."""
        src, dst = edge.in_node, edge.out_node
        c = self.new_node(src.bases + dst.bases[edge.weight:])
        if src is not dst:
            c.count = src.count + dst.count
            c.prevalence = src.prevalence + dst.prevalence
        else:
            c.count = src.count
            c.prevalence = src.prevalence
        c.norm = src.norm + dst.norm
        if c.norm == 0:
            c.copy_count = src.copy_count + dst.copy_count
        else:
            c.copy_count = (src.copy_count * src.norm + dst.copy_count * dst.norm) / c.norm
        if src is dst:
            edge.destroy()
            for e in list(src.out_edges):
                ne = self.link(c, e.out_node, e.weight)
                ne.copy_count = e.copy_count
                e.destroy()
            for e in list(src.in_edges):
                ne = self.link(e.in_node, c, e.weight)
                ne.copy_count = e.copy_count
                e.destroy()
            c.copy_count = src.copy_count / 2.0
            c.norm = src.norm
            c.reads = src.reads
            src.reads = []
            self.destroy_node(src)
            return c
        for e in list(src.in_edges):
            self.link(e.in_node, c, e.weight)
            e.destroy()
        for e in list(dst.out_edges):
            self.link(c, e.out_node, e.weight)
            e.destroy()
        shift = len(src.bases) - edge.weight
        src_set = set((r.rid, i) for r, i in src.reads)
        src_list = sorted(set((r.rid, i, r) for r, i in src.reads), key=lambda t: (t[0], t[1]))   # P1
        dst_reads = [(r, i - shift) for r, i in dst.reads]
        dst_reads = [(r, i) for r, i in dst_reads if (r.rid, i) not in src_set]
        c.reads = [(r, i) for _, i, r in src_list] + dst_reads
        src.reads, dst.reads = [], []
        edge.destroy()
        self.destroy_node(src)
        self.destroy_node(dst)
        return c

    def local_condense_edge(self, edge):
        """mbgraph.py:259-271."""
        if edge.in_node is None:
            return
        if len(edge.in_node.out_edges) > 1 or len(edge.out_node.in_edges) > 1:
            return
        c = self.condense_edge(edge)
        for e in c.in_edges + c.out_edges:
            self.local_condense_edge(e)

    def local_condense_node(self, n):
        """mbgraph.py:1315-1321."""
        if len(n.out_edges) == 1:
            self.local_condense_edge(n.out_edges[0])
        if len(n.in_edges) == 1:
            self.local_condense_edge(n.in_edges[0])

    def condense_all(self):
        """mbgraph.py:479-498.  The reference iterates the live Node.nodes list, so nodes
        created by condensing are visited later in the same pass."""
        i = 0
        while i < len(self.nodes):
            n = self.nodes[i]
            i += 1
            if len(n.out_edges) != 1:
                continue
            assert not n.destroyed
            e = n.out_edges[0]
            if len(e.out_node.in_edges) == 1 and n is not e.out_node:
                self.condense_edge(e)
        self.remove_destroyed()

    # ---------------------------------------------------------------- error pruning
    def is_suspicious(self, n):
        """mbgraph.py:1185-1226."""
        indeg, outdeg = len(n.in_edges), len(n.out_edges)
        if len(n.bases) <= self.SIZE_THRESHOLD and (indeg == 0 or outdeg == 0):
            return True
        if n.average_prevalence() >= self.PREVALENCE_THRESHOLD:
            return False
        if indeg == 0 or outdeg == 0:
            return True
        preds = n.predecessors()
        if preds:
            if float(sum(len(p.out_edges) for p in preds)) / len(preds) < 2:
                return False
        succs = n.successors()
        if succs:
            if float(sum(len(s.in_edges) for s in succs)) / len(succs) < 2:
                return False
        return True

    def destroy_some_suspicious(self):
        """mbgraph.py:1292-1313."""
        sus = [n for n in self.nodes if self.is_suspicious(n)]
        if not sus:
            return False
        sus.sort(key=lambda n: n.average_prevalence())
        for n in sus:
            if n.destroyed:
                continue
            adj = [e.in_node for e in n.in_edges] + [e.out_node for e in n.out_edges]
            self.full_destroy(n)
            for a in adj:
                self.local_condense_node(a)
        self.remove_destroyed()
        return True

    def destroy_suspicious(self):
        while self.destroy_some_suspicious():
            pass

    def similar(self, a, b):
        """mbgraph.py:1271-1289."""
        if a.destroyed or b.destroyed:
            return False
        if len(a.bases) != len(b.bases):
            return False
        mism = sum(1 for x, y in zip(a.bases, b.bases) if x != y)
        if float(mism) / max(len(a.bases), 1) >= self.HAMMING_FRACTION:
            return False
        if set(id(x) for x in a.successors()) != set(id(x) for x in b.successors()):
            return False
        if set(id(x) for x in a.predecessors()) != set(id(x) for x in b.predecessors()):
            return False
        return True

    def collapse(self, a, b):
        """mbgraph.py:1246-1256."""
        if a.prevalence < b.prevalence:
            a, b = b, a
        a.prevalence += b.prevalence
        self.full_destroy(b)

    def collapse_out(self, n):
        """mbgraph.py:1258-1269."""
        succ = [s for s in n.successors() if not s.destroyed]
        for i in range(len(succ)):
            for j in range(i + 1, len(succ)):
                if self.similar(succ[i], succ[j]):
                    self.collapse(succ[i], succ[j])
                    return True
        return False

    def collapse_all(self):
        """mbgraph.py:1229-1244."""
        while True:
            collapsed = False
            for n in self.nodes:
                if self.collapse_out(n):
                    collapsed = True
            self.remove_destroyed()
            if not collapsed:
                return

    # ---------------------------------------------------------------- bridging
    def find_bridging_reads(self):
        """mbgraph.py:88-111."""
        K = self.K
        starts = {}
        for n in self.nodes:
            if n.is_xnode():
                starts.setdefault(n.bases[:K], []).append(n)
        for bases, read in self.reads.items():
            for start in range(1, len(bases) - K):
                xs = starts.get(bases[start:start + K])
                if xs:
                    for x in xs:
                        if read.bridges(x, start):
                            x.reads.append((read, start))

    def refresh_bridging_reads(self, n):
        """mbgraph.py:450-476."""
        lb = len(n.bases)
        reads = [(r, i) for r, i in n.reads
                 if i > 0 and len(r.bases) > i + lb and r.bases[i:i + lb] == n.bases]
        reads = [(r, i) for _, i, r in sorted(set((r.rid, i, r) for r, i in reads), key=lambda t: (t[0], t[1]))]  # P1
        real = []
        for r, i in reads:
            bin_, bout = False, False
            for e in n.in_edges:
                if r.bases[i - 1] == e.in_node.bases[len(e.in_node.bases) - e.weight - 1]:
                    bin_ = True
            for e in n.out_edges:
                if r.bases[i + lb] == e.out_node.bases[e.weight]:
                    bout = True
            if bin_ and bout:
                real.append((r, i))
        n.reads = real

    def is_bridged_xnode(self, n):
        """mbgraph.py:514-529."""
        self.refresh_bridging_reads(n)
        inb, outb = set(), set()
        for r, i in n.reads:
            inb.add(r.bases[i - 1])
            outb.add(r.bases[i + len(n.bases)])
        bi = len(n.in_edges) - len(inb)
        bo = len(n.out_edges) - len(outb)
        return (bi == 0 and bo == 0) or (bi == 1 and bo == 1)

    def extend_forward(self, e):
        """mbgraph.py:273-285."""
        v, q = e.in_node, e.out_node
        w = self.new_node(v.bases + q.bases[e.weight])
        self.link(w, q, e.weight + 1)
        for r, i in v.reads:
            if r.bridges(w, i + 1):
                w.reads.append((r, i + 1))
        w.bridged = False
        return w

    def extend_back(self, e):
        """mbgraph.py:287-299."""
        p, v = e.in_node, e.out_node
        u = self.new_node(p.bases[-e.weight - 1] + v.bases)
        self.link(p, u, e.weight + 1)
        for r, i in v.reads:
            if r.bridges(u, i - 1):
                u.reads.append((r, i - 1))
        u.bridged = False
        return u

    def bridging_step(self, node):
        """mbgraph.py:552-628."""
        self.refresh_bridging_reads(node)
        assert len(node.reads) > 0
        assert len(node.in_edges) >= 2 and len(node.out_edges) >= 2
        u_list, w_list = [], []
        v_back, v_forward, loop_w = None, None, None
        in_edges, out_edges = list(node.in_edges), list(node.out_edges)
        for e in in_edges:
            u = self.extend_back(e)
            if e.in_node is node:
                v_back, loop_w = u, e.weight
            u_list.append(u)
        for e in out_edges:
            w = self.extend_forward(e)
            if e.out_node is node:
                v_forward = w
            w_list.append(w)
        for e in list(node.in_edges):
            e.destroy()
        for e in list(node.out_edges):
            e.destroy()
        if v_back is not None:
            assert v_forward is not None
            self.link(v_forward, v_back, loop_w + 2)
        links = {}
        for n in u_list + w_list:
            links[id(n)] = 0
        lb = len(node.bases)
        for r, i in list(node.reads):
            bu = r.bases[i - 1:i + lb]
            mu = [u for u in u_list if u.bases == bu]
            bw = r.bases[i:i + lb + 1]
            mw = [w for w in w_list if w.bases == bw]
            if len(mu) != 1 or len(mw) != 1:
                continue
            u, w = mu[0], mw[0]
            u.reads.append((r, i - 1))
            w.reads.append((r, i))
            if not u.precedes(w):
                self.link(u, w, lb)
                u.bridged = True
                w.bridged = True
                links[id(u)] += 1
                links[id(w)] += 1
        node.reads = []
        ub_u = [u for u in u_list if not u.bridged]
        ub_w = [w for w in w_list if not w.bridged]
        if len(ub_u) == 1 and len(ub_w) == 1:
            u, w = ub_u[0], ub_w[0]
            self.link(u, w, lb)
            links[id(u)] += 1
            links[id(w)] += 1
        else:
            assert len(ub_u) + len(ub_w) == 0
        link_count = sum(links.values())
        for n in u_list + w_list:
            n.prevalence = (float(links[id(n)]) / link_count) * node.prevalence
        for n in u_list + w_list:
            for e in n.in_edges + n.out_edges:
                self.local_condense_edge(e)
        self.destroy_node(node)

    def bridge_all(self):
        """mbgraph.py:537-550 (bridged_xnodes() is a generator consumed by list())."""
        while True:
            to_bridge = [n for n in self.nodes if n.is_xnode() and self.is_bridged_xnode(n)]
            for n in to_bridge:
                self.bridging_step(n)
            self.log.append("Bridged %d nodes" % len(to_bridge))
            self.remove_destroyed()
            if not to_bridge:
                return

    # ---------------------------------------------------------------- copy counts / cycles
    def find_approximate_copy_counts(self):
        """mbgraph.py:750-767."""
        self.known_paths = set()
        for n in self.nodes:
            n.norm = len(n.bases) - self.K + 1
            n.copy_count = float(n.prevalence) / n.norm
        for n in self.nodes:
            for e in n.out_edges:
                norm = max(self.L - e.weight - 1, 0)
                a, b = e.in_node, e.out_node
                cnt = a.copy_count * a.norm + b.copy_count * b.norm
                e.copy_count = 0 if norm == 0 else 0.5 * cnt / norm

    def disregard_loops(self):
        """mbgraph.py:1324-1331."""
        for n in self.nodes:
            if any(e.out_node is n for e in n.out_edges):
                n.norm = 0
                n.copy_count = 0

    def reachable_cycle(self, n, no_cycles, traversed):
        """mbgraph.py:903-928."""
        traversed = list(traversed)
        traversed.append(n)
        for m in [e.out_node for e in n.out_edges]:
            if any(m is t for t in traversed):
                cyc = traversed + [m]
                k = next(i for i, t in enumerate(cyc) if t is m)
                return cyc[k:]
            if id(m) in no_cycles:
                continue
            c = self.reachable_cycle(m, no_cycles, traversed)
            if c:
                return c
        no_cycles.add(id(n))
        return None

    def find_cycle(self, no_cycles):
        """mbgraph.py:950-960."""
        for n in self.nodes:
            if id(n) not in no_cycles:
                c = self.reachable_cycle(n, no_cycles, [])
                if c:
                    return c
        return None

    def break_cycles(self):
        """mbgraph.py:1133-1161 with dfs=False; break_cycle :1040-1049 with CYCLE_DESTROY=True:
        full_destroy the second element of the found cycle list."""
        no_cycles = set()
        c = self.find_cycle(no_cycles)
        while c is not None:
            self.full_destroy(c[1:][0])
            c = self.find_cycle(no_cycles)
        self.remove_destroyed()
        self.condense_all()
        assert self.find_cycle(set()) is None

    # ---------------------------------------------------------------- reads on graph
    @staticmethod
    def compare(a, b):
        n = min(len(a), len(b))
        return a[:n] == b[:n]

    def search_sequence(self, seq, node, i, max_hops):
        """mbgraph.py:1416-1436."""
        nl = len(node.bases) - i
        if max_hops <= 0:
            return [[node]]
        if len(seq) <= nl:
            return [[node]]
        seq = seq[nl:]
        es = [e for e in node.out_edges if self.compare(seq, e.out_node.bases[e.weight:])]
        if not es:
            return []
        paths = []
        for e in es:
            for p in self.search_sequence(seq, e.out_node, e.weight, max_hops - 1):
                paths.append([node] + p)
        return paths

    def find_known_paths(self):
        """mbgraph.py:1355-1388."""
        K = self.K
        self.known_paths = set()
        kmers = {}
        for n in self.nodes:
            for i in range(len(n.bases) - K + 1):
                kmers.setdefault(n.bases[i:i + K], []).append((n, i))
        cnt = 0
        for bases, read in self.reads.items():
            sk, ek = bases[:K], bases[-K:]
            if sk not in kmers or ek not in kmers:
                continue
            for sn, si in kmers[sk]:
                if not self.compare(bases, sn.bases[si:]):
                    continue
                for path in self.search_sequence(bases, sn, si, 30):
                    read.nodes = path
                    for j in range(len(path) - 1):
                        key = (id(path[j]), id(path[j + 1]))
                        self.known_edges[key] = self.known_edges.get(key, 0) + read.copy_count
                    if len(path) > 2:
                        self.known_paths.add(tuple(path))
                        cnt += 1
        self.log.append("No of known paths:%d" % cnt)

    def find_copy_counts(self):
        """mbgraph.py:735-746."""
        for n in self.nodes:
            tot = 0
            for e in n.out_edges:
                ec = self.known_edges.get((id(e.in_node), id(e.out_node)), 0)
                tot += ec
                e.copy_count = ec / max(self.L - e.weight - 1, 1)
            n.copy_count = tot

    def mate_search(self, n, goal, max_length, min_length, max_hops):
        """mbgraph.py:860-880."""
        if max_length <= 0 or max_hops <= 0:
            return []
        if n is goal and min_length <= 1:
            return [[goal]]
        paths = []
        for e in n.out_edges:
            nl = len(n.bases) - e.weight
            for p in self.mate_search(e.out_node, goal, max_length - nl, min_length - nl, max_hops - 1):
                paths.append([n] + list(p))
        return paths

    def find_mate_path(self, n, start_base, goal, end_base):
        """mbgraph.py:839-858."""
        fringe = end_base + (len(n.bases) - start_base)
        min_l = self.MATE_PAIR_MIN_LENGTH - fringe
        max_l = self.MATE_PAIR_LENGTH - fringe
        paths = []
        for e in n.out_edges:
            for p in self.mate_search(e.out_node, goal, max_l + e.weight, min_l + e.weight, self.MATE_PAIR_MAX_HOPS):
                paths.append([n] + p)
        return paths

    def find_mate_pairs(self):
        """mbgraph.py:114-160."""
        pairs = {}
        for bases, r in self.reads.items():
            if r.mate_pair == 1 and r.nodes and r.mate.nodes:
                a, b = r.nodes[-1], r.mate.nodes[0]
                if a is b:
                    continue
                if any(s is b for s in a.successors()):
                    continue
                pairs[(id(a), id(b))] = (a, b)
        n_mp = 0
        for a, b in pairs.values():
            paths = self.find_mate_path(a, len(a.bases) - 1, b, 0)
            if len(paths) == 1 and len(paths[0]) > 2:
                n_mp += 1
                self.known_paths.add(tuple(paths[0]))
                self.mate_paths.add(tuple(paths[0]))
        self.log.append("No of mate paths: %d" % n_mp)

    # ---------------------------------------------------------------- pipeline + output
    def run(self, error_correction=True):
        """multibridging.py:209-269."""
        self.condense_all()
        self.log.append("%d nodes after condensing." % len(self.nodes))
        if error_correction:
            self.destroy_suspicious()
            self.log.append("%d nodes after destroying suspicious nodes." % len(self.nodes))
            self.collapse_all()
            self.log.append("%d nodes after collapsing similar nodes." % len(self.nodes))
        self.find_bridging_reads()
        self.bridge_all()
        self.condense_all()
        self.log.append("%d nodes after bridging." % len(self.nodes))
        self.find_approximate_copy_counts()
        self.disregard_loops()
        self.condense_all()
        self.remove_destroyed()
        self.break_cycles()
        self.find_approximate_copy_counts()
        self.find_known_paths()
        self.find_copy_counts()
        self.find_mate_pairs()
        self.log.append("%d final nodes." % len(self.nodes))

    def add_component(self, src):
        """mbgraph.py:691-709.  Returns (node list in creation order, edge list)."""
        seen, edges = {}, {}
        queue = [src]
        while queue:
            n = queue.pop()
            if id(n) in seen:
                continue
            seen[id(n)] = n
            for e in n.out_edges:
                edges[id(e)] = e
            for e in n.out_edges:
                queue.append(e.out_node)
            for e in n.in_edges:
                queue.append(e.in_node)
        return sorted(seen.values(), key=lambda n: n.nid), list(edges.values())

    def topological_sort(self, nodes):
        """mbgraph.py:711-732 (P2: `nodes` scanned in creation order)."""
        added, out = set(), []
        fringe = [n for n in nodes if len(n.in_edges) == 0]
        while fringe:
            v = fringe.pop()
            if id(v) in added:
                continue
            added.add(id(v))
            out.append(v)
            for n in [e.out_node for e in v.out_edges]:
                if all(id(p) in added for p in n.predecessors()):
                    fringe.append(n)
        return out

    def output_components(self):
        """multibridging.py:271-325.  Returns (single_rows, components) where
        components = [dict(nodes=[(id,bases,cc,norm)], edges=[(in,out,w,cc,norm)], paths=[[ids]])]."""
        singles, comps = [], []
        by_start = {}
        for p in self.known_paths:
            by_start.setdefault(id(p[0]), []).append(p)
        for src in self.nodes:
            if src.destroyed:
                continue
            cn, ce = self.add_component(src)
            cn = self.topological_sort(cn)
            if len(cn) == 1:
                src.hash = -1
                singles.append((-1, src.bases, src.copy_count, src.norm))
                src.destroyed = True
                continue
            for h, n in enumerate(cn):
                n.hash = h
                n.destroyed = True
            nodes = [(n.hash, n.bases, n.copy_count, n.norm) for n in cn]
            paths = []
            for n in cn:
                ps = by_start.get(id(n))
                if ps:
                    paths.extend(sorted([[x.hash for x in p] for p in ps]))
            edges = sorted([(e.in_node.hash, e.out_node.hash, e.weight, e.copy_count, max(self.L - e.weight - 1, 0))
                            for e in ce if e.copy_count > 0], key=lambda t: (t[0], t[1], t[2]))
            comps.append({"nodes": nodes, "edges": edges, "paths": paths})
        return singles, comps


def run_partition(k1mer_rows, reads, K, paired=False):
    """multibridging.main (multibridging.py:327-400) for one partition, in memory.
    reads = [list] (SE) or [list1, list2] (PE).  Read.L = len(first read) (:197-204)."""
    L = len(reads[0][0]) if reads[0] else 0
    g = MBGraph(K, L)
    g.load_k1mers(k1mer_rows)
    if paired:
        g.load_mated_reads(reads[0], reads[1])
    else:
        g.load_reads(reads[0])
    g.run(True)
    singles, comps = g.output_components()
    return g, singles, comps


def canonical(singles, comps):
    """ID-free canonical form, same layout as tests/golden/ref_harness.canonical_graph."""
    out = {"single_nodes": sorted([[b, float(cc), float(nm)] for _, b, cc, nm in singles]),
           "nodes": [], "edges": [], "paths": []}
    for c in comps:
        id2b = {h: b for h, b, _, _ in c["nodes"]}
        out["nodes"] += [[b, float(cc), float(nm)] for _, b, cc, nm in c["nodes"]]
        out["edges"] += [[id2b[a], id2b[b], w, float(cc), float(nm)] for a, b, w, cc, nm in c["edges"]]
        out["paths"] += [[id2b[x] for x in p] for p in c["paths"]]
    for k in ("nodes", "edges", "paths"):
        out[k].sort()
    return out
