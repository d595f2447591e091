"""Multibridged de-Bruijn graph of one partition -- host mirror of the reference's
multibridging.py + mbgraph.py (rows a12-a24), array/index based.

Nodes and edges are integer ids into parallel lists (no object graph, no class-level globals:
re-entrant, one Graph per partition).  Where the reference iterates Python sets of objects
(address order) the order is pinned exactly as in DESIGN.md "pinned orders" (P1-P3).

The read-on-graph scans (find_bridging_reads mbgraph.py:88-111 and known_paths :1355-1388) take
their K-mer seed hits from `seed_hits` callbacks so the HIP seed-lookup kernel can supply them.
"""
import sys


class Graph(object):
    MATE_PAIR_LENGTH = 300      # mbgraph.py:18
    MATE_PAIR_MIN_LENGTH = 0
    MATE_PAIR_MAX_HOPS = 7      # mbgraph.py:20
    PREVALENCE_THRESHOLD = 1    # mbgraph.py:324
    HAMMING_FRACTION = 0.1      # mbgraph.py:326

    def __init__(self, K, L):
        self.K, self.L = K, L
        self.SIZE_THRESHOLD = L                     # multibridging.py:203
        # node columns
        self.bases, self.ine, self.oute = [], [], []
        self.norm, self.cc, self.prev, self.cnt = [], [], [], []
        self.dead, self.nreads, self.bridged, self.hash = [], [], [], []
        self.order = []                             # Node.nodes
        # edge columns (src == -1: destroyed)
        self.es, self.ed, self.ew, self.ecc = [], [], [], []
        # reads
        self.rbases, self.rcc, self.rmate, self.rmp, self.rnodes = [], [], [], [], []
        self.rindex = {}
        self.known_paths = set()
        self.known_edges = {}
        self.log = []

    # ------------------------------------------------------------------ primitives
    def new_node(self, bases):
        """mbgraph.py:328-341."""
        n = len(self.bases)
        self.bases.append(bases)
        self.ine.append([])
        self.oute.append([])
        self.norm.append(1.0)
        self.cc.append(0.0)
        self.prev.append(0.0)
        self.cnt.append(1.0)
        self.dead.append(False)
        self.nreads.append([])
        self.bridged.append(None)
        self.hash.append(None)
        self.order.append(n)
        return n

    def link(self, a, b, w):
        """mbgraph.py:169-176."""
        e = len(self.es)
        self.es.append(a)
        self.ed.append(b)
        self.ew.append(w)
        self.ecc.append(0.0)
        self.oute[a].append(e)
        self.ine[b].append(e)
        return e

    def kill_edge(self, e):
        """mbgraph.py:178-182."""
        self.oute[self.es[e]].remove(e)
        self.ine[self.ed[e]].remove(e)
        self.es[e] = -1
        self.ed[e] = -1

    def kill_node(self, n):
        """Node.destroy(pop=False), mbgraph.py:423-435."""
        assert not self.ine[n] and not self.oute[n] and not self.nreads[n]
        self.dead[n] = True

    def full_destroy(self, n):
        """mbgraph.py:436-446."""
        for e in list(self.ine[n]):
            self.kill_edge(e)
        for e in list(self.oute[n]):
            self.kill_edge(e)
        self.nreads[n] = []
        self.kill_node(n)

    def remove_destroyed(self):
        self.order = [n for n in self.order if not self.dead[n]]

    def succ(self, n):
        return [self.ed[e] for e in self.oute[n]]

    def pred(self, n):
        return [self.es[e] for e in self.ine[n]]

    def is_xnode(self, n):
        return len(self.ine[n]) >= 2 and len(self.oute[n]) >= 2

    def avg_prev(self, n):
        return self.prev[n] / self.cnt[n]

    # ------------------------------------------------------------------ loading
    def load_k1mers(self, rows):
        """load_single_jellyfish, multibridging.py:145-172."""
        K = self.K
        idx = {}
        for km, prevalence in rows:
            assert K == len(km) - 1
            a, b = km[:-1], km[1:]
            na = idx.get(a)
            if na is None:
                na = idx[a] = self.new_node(a)
            nb = idx.get(b)
            if nb is None:
                nb = idx[b] = self.new_node(b)
            e = self.link(na, nb, K - 1)
            self.ecc[e] = round(float(prevalence))
        for n in self.order:
            self.prev[n] = sum(self.ew[e] for e in self.oute[n])      # quirk :171-172

    def add_read(self, bases, cc=1.0):
        """Read.add_read, mbgraph.py:44-62."""
        r = self.rindex.get(bases)
        if r is not None:
            self.rcc[r] += cc
            return r
        r = len(self.rbases)
        self.rindex[bases] = r
        self.rbases.append(bases)
        self.rcc.append(cc)
        self.rmate.append(None)
        self.rmp.append(None)
        self.rnodes.append(None)
        return r

    def load_reads(self, reads):
        """multibridging.py:22-30."""
        cutoff = len(self.order) * 10
        for i, s in enumerate(reads):
            if i > cutoff:
                break
            self.add_read(s.upper())

    def load_mated_reads(self, reads1, reads2):
        """multibridging.py:68-97."""
        cutoff = len(self.order) * 10
        for i, (a, b) in enumerate(zip(reads1, reads2)):
            if i > cutoff:
                break
            r1 = self.add_read(a.upper().strip())
            r2 = self.add_read(b.upper().strip())
            self.rmp[r1] = 1
            self.rmp[r2] = 2
            self.rmate[r1] = r2
            self.rmate[r2] = r1

    # ------------------------------------------------------------------ condensing
    def condense(self, e):
        """Edge.condense(pop=False), mbgraph.py:184-257."""
        s, d, w = self.es[e], self.ed[e], self.ew[e]
        c = self.new_node(self.bases[s] + self.bases[d][w:])
        if s != d:
            self.cnt[c] = self.cnt[s] + self.cnt[d]
            self.prev[c] = self.prev[s] + self.prev[d]
        else:
            self.cnt[c] = self.cnt[s]
            self.prev[c] = self.prev[s]
        self.norm[c] = self.norm[s] + self.norm[d]
        if self.norm[c] == 0:
            self.cc[c] = self.cc[s] + self.cc[d]
        else:
            self.cc[c] = (self.cc[s] * self.norm[s] + self.cc[d] * self.norm[d]) / self.norm[c]
        if s == d:
            self.kill_edge(e)
            for x in list(self.oute[s]):
                ne = self.link(c, self.ed[x], self.ew[x])
                self.ecc[ne] = self.ecc[x]
                self.kill_edge(x)
            for x in list(self.ine[s]):
                ne = self.link(self.es[x], c, self.ew[x])
                self.ecc[ne] = self.ecc[x]
                self.kill_edge(x)
            self.cc[c] = self.cc[s] / 2.0
            self.norm[c] = self.norm[s]
            self.nreads[c] = self.nreads[s]
            self.nreads[s] = []
            self.kill_node(s)
            return c
        for x in list(self.ine[s]):
            self.link(self.es[x], c, self.ew[x])
            self.kill_edge(x)
        for x in list(self.oute[d]):
            self.link(c, self.ed[x], self.ew[x])
            self.kill_edge(x)
        shift = len(self.bases[s]) - w
        sset = set(self.nreads[s])
        dreads = [(r, i - shift) for r, i in self.nreads[d]]
        self.nreads[c] = sorted(sset) + [x for x in dreads if x not in sset]      # P1
        self.nreads[s] = []
        self.nreads[d] = []
        self.kill_edge(e)
        self.kill_node(s)
        self.kill_node(d)
        return c

    def local_condense_edge(self, e):
        """Edge.local_condense, mbgraph.py:259-271."""
        if self.es[e] < 0:
            return
        if len(self.oute[self.es[e]]) > 1 or len(self.ine[self.ed[e]]) > 1:
            return
        c = self.condense(e)
        for x in self.ine[c] + self.oute[c]:
            self.local_condense_edge(x)

    def local_condense_node(self, n):
        """Node.local_condense, mbgraph.py:1315-1321."""
        if len(self.oute[n]) == 1:
            self.local_condense_edge(self.oute[n][0])
        if len(self.ine[n]) == 1:
            self.local_condense_edge(self.ine[n][0])

    def condense_all(self):
        """mbgraph.py:479-498 (iterates the live list: nodes created meanwhile are visited too)."""
        i = 0
        order = self.order
        while i < len(order):
            n = order[i]
            i += 1
            if len(self.oute[n]) != 1:
                continue
            e = self.oute[n][0]
            d = self.ed[e]
            if len(self.ine[d]) == 1 and n != d:
                self.condense(e)
        self.remove_destroyed()

    # ------------------------------------------------------------------ error pruning
    def is_suspicious(self, n):
        """mbgraph.py:1185-1226."""
        ni, no = len(self.ine[n]), len(self.oute[n])
        if len(self.bases[n]) <= self.SIZE_THRESHOLD and (ni == 0 or no == 0):
            return True
        if self.avg_prev(n) >= self.PREVALENCE_THRESHOLD:
            return False
        if ni == 0 or no == 0:
            return True
        ps = self.pred(n)
        if ps and float(sum(len(self.oute[p]) for p in ps)) / len(ps) < 2:
            return False
        ss = self.succ(n)
        if ss and float(sum(len(self.ine[s]) for s in ss)) / len(ss) < 2:
            return False
        return True

    def destroy_suspicious(self):
        """mbgraph.py:1169-1172, 1292-1313."""
        while True:
            sus = [n for n in self.order if self.is_suspicious(n)]
            if not sus:
                return
            sus.sort(key=self.avg_prev)
            for n in sus:
                if self.dead[n]:
                    continue
                adj = self.pred(n) + self.succ(n)
                self.full_destroy(n)
                for a in adj:
                    self.local_condense_node(a)
            self.remove_destroyed()

    def similar(self, a, b):
        """mbgraph.py:1271-1289."""
        if self.dead[a] or self.dead[b]:
            return False
        x, y = self.bases[a], self.bases[b]
        if len(x) != len(y):
            return False
        mism = sum(1 for p, q in zip(x, y) if p != q)
        if float(mism) / max(len(x), 1) >= self.HAMMING_FRACTION:
            return False
        return set(self.succ(a)) == set(self.succ(b)) and set(self.pred(a)) == set(self.pred(b))

    def collapse_all(self):
        """mbgraph.py:1229-1269."""
        while True:
            collapsed = False
            for n in self.order:
                ss = [s for s in self.succ(n) if not self.dead[s]]
                done = False
                for i in range(len(ss)):
                    for j in range(i + 1, len(ss)):
                        if self.similar(ss[i], ss[j]):
                            a, b = ss[i], ss[j]
                            if self.prev[a] < self.prev[b]:
                                a, b = b, a
                            self.prev[a] += self.prev[b]
                            self.full_destroy(b)
                            collapsed = done = True
                            break
                    if done:
                        break
            self.remove_destroyed()
            if not collapsed:
                return

    # ------------------------------------------------------------------ bridging
    def read_bridges(self, r, n, index):
        """Read.bridges, mbgraph.py:77-85."""
        rb, nb = self.rbases[r], self.bases[n]
        if index <= 0 or len(rb) <= index + len(nb):
            return False
        return rb[index:index + len(nb)] == nb

    def find_bridging_reads(self, seed_hits=None):
        """mbgraph.py:88-111.  seed_hits(starts) -> iterable of (read id, start, [xnodes]) in
        (read, start) order; default = host scan."""
        K = self.K
        starts = {}
        for n in self.order:
            if self.is_xnode(n):
                starts.setdefault(self.bases[n][:K], []).append(n)
        if seed_hits is None:
            def seed_hits(st):
                for r, rb in enumerate(self.rbases):
                    for s in range(1, len(rb) - K):
                        xs = st.get(rb[s:s + K])
                        if xs:
                            yield r, s, xs
        for r, s, xs in seed_hits(starts):
            for x in xs:
                if self.read_bridges(r, x, s):
                    self.nreads[x].append((r, s))

    def refresh_bridging_reads(self, n):
        """mbgraph.py:450-476."""
        nb = self.bases[n]
        lb = len(nb)
        rs = sorted(set((r, i) for r, i in self.nreads[n]
                        if i > 0 and len(self.rbases[r]) > i + lb and self.rbases[r][i:i + lb] == nb))   # P1
        real = []
        for r, i in rs:
            rb = self.rbases[r]
            bi = any(rb[i - 1] == self.bases[self.es[e]][len(self.bases[self.es[e]]) - self.ew[e] - 1] for e in self.ine[n])
            bo = any(rb[i + lb] == self.bases[self.ed[e]][self.ew[e]] for e in self.oute[n])
            if bi and bo:
                real.append((r, i))
        self.nreads[n] = real

    def is_bridged_xnode(self, n):
        """mbgraph.py:514-529."""
        self.refresh_bridging_reads(n)
        lb = len(self.bases[n])
        inb = set(self.rbases[r][i - 1] for r, i in self.nreads[n])
        outb = set(self.rbases[r][i + lb] for r, i in self.nreads[n])
        bi = len(self.ine[n]) - len(inb)
        bo = len(self.oute[n]) - len(outb)
        return (bi == 0 and bo == 0) or (bi == 1 and bo == 1)

    def bridging_step(self, node):
        """mbgraph.py:552-628 with extend_back/forward :273-299."""
        self.refresh_bridging_reads(node)
        assert len(self.nreads[node]) > 0
        assert len(self.ine[node]) >= 2 and len(self.oute[node]) >= 2
        nb = self.bases[node]
        lb = len(nb)
        u_list, w_list = [], []
        v_back = v_forward = loop_w = None
        for e in list(self.ine[node]):
            p, w = self.es[e], self.ew[e]
            u = self.new_node(self.bases[p][-w - 1] + nb)
            self.link(p, u, w + 1)
            for r, i in self.nreads[node]:
                if self.read_bridges(r, u, i - 1):
                    self.nreads[u].append((r, i - 1))
            self.bridged[u] = False
            if p == node:
                v_back, loop_w = u, w
            u_list.append(u)
        for e in list(self.oute[node]):
            q, w = self.ed[e], self.ew[e]
            x = self.new_node(nb + self.bases[q][w])
            self.link(x, q, w + 1)
            for r, i in self.nreads[node]:
                if self.read_bridges(r, x, i + 1):
                    self.nreads[x].append((r, i + 1))
            self.bridged[x] = False
            if q == node:
                v_forward = x
            w_list.append(x)
        for e in list(self.ine[node]):
            self.kill_edge(e)
        for e in list(self.oute[node]):
            self.kill_edge(e)
        if v_back is not None:
            assert v_forward is not None
            self.link(v_forward, v_back, loop_w + 2)
        links = {n: 0 for n in u_list + w_list}
        for r, i in list(self.nreads[node]):
            rb = self.rbases[r]
            bu, bw = rb[i - 1:i + lb], rb[i:i + lb + 1]
            mu = [u for u in u_list if self.bases[u] == bu]
            mw = [x for x in w_list if self.bases[x] == bw]
            if len(mu) != 1 or len(mw) != 1:
                continue
            u, x = mu[0], mw[0]
            self.nreads[u].append((r, i - 1))
            self.nreads[x].append((r, i))
            if x not in self.succ(u):
                self.link(u, x, lb)
                self.bridged[u] = True
                self.bridged[x] = True
                links[u] += 1
                links[x] += 1
        self.nreads[node] = []
        ub_u = [u for u in u_list if not self.bridged[u]]
        ub_w = [x for x in w_list if not self.bridged[x]]
        if len(ub_u) == 1 and len(ub_w) == 1:
            self.link(ub_u[0], ub_w[0], lb)
            links[ub_u[0]] += 1
            links[ub_w[0]] += 1
        else:
            assert len(ub_u) + len(ub_w) == 0
        link_count = sum(links.values())
        for n in u_list + w_list:
            self.prev[n] = (float(links[n]) / link_count) * self.prev[node]
        for n in u_list + w_list:
            for e in self.ine[n] + self.oute[n]:
                self.local_condense_edge(e)
        self.kill_node(node)

    def bridge_all(self):
        """mbgraph.py:537-550."""
        while True:
            todo = [n for n in self.order if self.is_xnode(n) and self.is_bridged_xnode(n)]
            for n in todo:
                self.bridging_step(n)
            self.log.append("Bridged %d nodes" % len(todo))
            self.remove_destroyed()
            if not todo:
                return

    # ------------------------------------------------------------------ copy counts, cycles
    def find_approximate_copy_counts(self):
        """mbgraph.py:750-767."""
        self.known_paths = set()
        for n in self.order:
            self.norm[n] = len(self.bases[n]) - self.K + 1
            self.cc[n] = float(self.prev[n]) / self.norm[n]
        for n in self.order:
            for e in self.oute[n]:
                nm = max(self.L - self.ew[e] - 1, 0)
                a, b = self.es[e], self.ed[e]
                tot = self.cc[a] * self.norm[a] + self.cc[b] * self.norm[b]
                self.ecc[e] = 0 if nm == 0 else 0.5 * tot / nm

    def disregard_loops(self):
        """mbgraph.py:1324-1331."""
        for n in self.order:
            if n in self.succ(n):
                self.norm[n] = 0
                self.cc[n] = 0

    def _reachable_cycle(self, n, no_cycles, trav):
        """mbgraph.py:903-928."""
        trav = trav + [n]
        for m in self.succ(n):
            if m in trav:
                cyc = trav + [m]
                return cyc[cyc.index(m):]
            if m in no_cycles:
                continue
            c = self._reachable_cycle(m, no_cycles, trav)
            if c:
                return c
        no_cycles.add(n)
        return None

    def find_cycle(self, no_cycles):
        """mbgraph.py:950-960."""
        for n in self.order:
            if n not in no_cycles:
                c = self._reachable_cycle(n, no_cycles, [])
                if c:
                    return c
        return None

    def break_cycles(self):
        """mbgraph.py:1133-1161 (dfs=False) + break_cycle :1040-1049 (CYCLE_DESTROY=True)."""
        no_cycles = set()
        c = self.find_cycle(no_cycles)
        while c is not None:
            self.full_destroy(c[1])
            c = self.find_cycle(no_cycles)
        self.remove_destroyed()
        self.condense_all()
        assert self.find_cycle(set()) is None

    # ------------------------------------------------------------------ reads on the graph
    @staticmethod
    def _compare(a, b):
        n = min(len(a), len(b))
        return a[:n] == b[:n]

    def search_sequence(self, seq, node, i, max_hops):
        """mbgraph.py:1416-1436."""
        nl = len(self.bases[node]) - i
        if max_hops <= 0 or len(seq) <= nl:
            return [[node]]
        seq = seq[nl:]
        es = [e for e in self.oute[node] if self._compare(seq, self.bases[self.ed[e]][self.ew[e]:])]
        paths = []
        for e in es:
            for p in self.search_sequence(seq, self.ed[e], self.ew[e], max_hops - 1):
                paths.append([node] + p)
        return paths

    def find_known_paths(self, seed_hits=None):
        """known_paths(), mbgraph.py:1355-1388.  seed_hits(kmers) -> iterable of
        (read id, [(node, offset)...]) for reads whose first AND last K-mer are indexed."""
        K = self.K
        self.known_paths = set()
        kmers = {}
        for n in self.order:
            b = self.bases[n]
            for i in range(len(b) - K + 1):
                kmers.setdefault(b[i:i + K], []).append((n, i))
        if seed_hits is None:
            def seed_hits(km):
                for r, rb in enumerate(self.rbases):
                    st = km.get(rb[:K])
                    if st is not None and rb[-K:] in km:
                        yield r, st
        cnt = 0
        for r, st in seed_hits(kmers):
            rb = self.rbases[r]
            for sn, si in st:
                if not self._compare(rb, self.bases[sn][si:]):
                    continue
                for path in self.search_sequence(rb, sn, si, 30):
                    self.rnodes[r] = path
                    for j in range(len(path) - 1):
                        key = (path[j], path[j + 1])
                        self.known_edges[key] = self.known_edges.get(key, 0) + self.rcc[r]
                    if len(path) > 2:
                        self.known_paths.add(tuple(path))
                        cnt += 1
        self.log.append("No of known paths:%d" % cnt)

    def find_copy_counts(self):
        """mbgraph.py:735-746."""
        for n in self.order:
            tot = 0
            for e in self.oute[n]:
                ec = self.known_edges.get((self.es[e], self.ed[e]), 0)
                tot += ec
                self.ecc[e] = ec / max(self.L - self.ew[e] - 1, 1)
            self.cc[n] = tot

    def _mate_search(self, n, goal, max_len, min_len, hops):
        """mbgraph.py:860-880."""
        if max_len <= 0 or hops <= 0:
            return []
        if n == goal and min_len <= 1:
            return [[goal]]
        out = []
        for e in self.oute[n]:
            nl = len(self.bases[n]) - self.ew[e]
            for p in self._mate_search(self.ed[e], goal, max_len - nl, min_len - nl, hops - 1):
                out.append([n] + p)
        return out

    def find_mate_pairs(self):
        """mbgraph.py:114-160, find_mate_path :839-858."""
        pairs = {}
        for r in range(len(self.rbases)):
            if self.rmp[r] == 1 and self.rnodes[r] and self.rnodes[self.rmate[r]]:
                a, b = self.rnodes[r][-1], self.rnodes[self.rmate[r]][0]
                if a == b or b in self.succ(a):
                    continue
                pairs[(a, b)] = True
        nmp = 0
        for a, b in pairs:
            fringe = 0 + (len(self.bases[a]) - (len(self.bases[a]) - 1))
            min_l = self.MATE_PAIR_MIN_LENGTH - fringe
            max_l = self.MATE_PAIR_LENGTH - fringe
            paths = []
            for e in self.oute[a]:
                for p in self._mate_search(self.ed[e], b, max_l + self.ew[e], min_l + self.ew[e], self.MATE_PAIR_MAX_HOPS):
                    paths.append([a] + p)
            if len(paths) == 1 and len(paths[0]) > 2:
                nmp += 1
                self.known_paths.add(tuple(paths[0]))
        self.log.append("No of mate paths: %d" % nmp)

    # ------------------------------------------------------------------ pipeline + output
    def run(self, bridging_hits=None, path_hits=None):
        """multibridging.run, multibridging.py:209-269 (error_correction=True)."""
        self.condense_all()
        self.log.append("%d nodes after condensing." % len(self.order))
        self.destroy_suspicious()
        self.log.append("%d nodes after destroying suspicious nodes." % len(self.order))
        self.collapse_all()
        self.log.append("%d nodes after collapsing similar nodes." % len(self.order))
        self.find_bridging_reads(bridging_hits)
        self.bridge_all()
        self.condense_all()
        self.log.append("%d nodes after bridging." % len(self.order))
        self.find_approximate_copy_counts()
        self.disregard_loops()
        self.condense_all()
        self.remove_destroyed()
        self.break_cycles()
        self.find_approximate_copy_counts()
        self.find_known_paths(path_hits)
        self.find_copy_counts()
        self.find_mate_pairs()
        self.log.append("%d final nodes." % len(self.order))

    def output_components(self):
        """multibridging.output_components, multibridging.py:271-325; add_component
        mbgraph.py:691-709; topological_sort :711-732 (P2).  Returns (single_rows, components)."""
        singles, comps = [], []
        by_start = {}
        for p in self.known_paths:
            by_start.setdefault(p[0], []).append(p)
        for src in self.order:
            if self.dead[src]:
                continue
            seen, edges, queue = set(), {}, [src]
            while queue:
                n = queue.pop()
                if n in seen:
                    continue
                seen.add(n)
                for e in self.oute[n]:
                    edges[e] = True
                queue.extend(self.ed[e] for e in self.oute[n])
                queue.extend(self.es[e] for e in self.ine[n])
            cn = sorted(seen)
            added, topo = set(), []
            fringe = [n for n in cn if not self.ine[n]]
            while fringe:
                v = fringe.pop()
                if v in added:
                    continue
                added.add(v)
                topo.append(v)
                for n in self.succ(v):
                    if all(p in added for p in self.pred(n)):
                        fringe.append(n)
            if len(topo) == 1:
                self.hash[src] = -1
                singles.append((-1, self.bases[src], self.cc[src], self.norm[src]))
                self.dead[src] = True
                continue
            for h, n in enumerate(topo):
                self.hash[n] = h
                self.dead[n] = True
            nodes = [(self.hash[n], self.bases[n], self.cc[n], self.norm[n]) for n in topo]
            paths = []
            for n in topo:
                ps = by_start.get(n)
                if ps:
                    paths.extend(sorted([self.hash[x] for x in p] for p in ps))
            el = sorted(((self.hash[self.es[e]], self.hash[self.ed[e]], self.ew[e], self.ecc[e], max(self.L - self.ew[e] - 1, 0))
                         for e in edges if self.ecc[e] > 0), key=lambda t: (t[0], t[1], t[2]))
            comps.append({"nodes": nodes, "edges": el, "paths": paths})
        return singles, comps


def run_partition(k1mer_rows, reads, K, paired=False, hits_factory=None):
    """multibridging.main for one partition (multibridging.py:327-400), in memory.
    reads = [list] or [list1, list2]; Read.L = len(first read) (setup, :197-204).
    hits_factory(graph) -> (bridging_hits, path_hits) lets the caller plug the HIP seed kernels."""
    L = len(reads[0][0]) if reads[0] else -1
    g = Graph(K, L)
    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(max(lim, 10000))          # multibridging.py:329
    try:
        g.load_k1mers(k1mer_rows)
        if paired:
            g.load_mated_reads(reads[0], reads[1])
        else:
            g.load_reads(reads[0])
        bh = ph = None
        if hits_factory is not None:
            bh, ph = hits_factory(g)
        g.run(bh, ph)
        singles, comps = g.output_components()
    finally:
        sys.setrecursionlimit(lim)
    return g, singles, comps


def canonical(singles, comps):
    out = {"single_nodes": sorted([[b, float(cc), float(nm)] for _, b, cc, nm in singles]), "nodes": [], "edges": [], "paths": []}
    for c in comps:
        id2b = {h: b for h, b, _, _ in c["nodes"]}
        out["nodes"] += [[b, float(cc), float(nm)] for _, b, cc, nm in c["nodes"]]
        out["edges"] += [[id2b[a], id2b[b], w, float(cc), float(nm)] for a, b, w, cc, nm in c["edges"]]
        out["paths"] += [[id2b[x] for x in p] for p in c["paths"]]
    for k in ("nodes", "edges", "paths"):
        out[k].sort()
    return out


def write_files(singles, comps, out_dir):
    """nodes/edges/paths{c}.txt + single_nodes.txt as multibridging.py:271-325 writes them."""
    import os
    with open(os.path.join(out_dir, "single_nodes.txt"), "w") as f:
        f.write("ID\tBases\tCopycount\tNormalization\n")
        for h, b, cc, nm in singles:
            f.write("%s\t%s\t%s\t%s\n" % (h, b, cc, nm))
    for c, comp in enumerate(comps):
        with open(os.path.join(out_dir, "nodes%d.txt" % c), "w") as f:
            f.write("ID\tBases\tCopycount\tNormalization\n")
            for h, b, cc, nm in comp["nodes"]:
                f.write("%s\t%s\t%s\t%s\n" % (h, b, cc, nm))
        with open(os.path.join(out_dir, "paths%d.txt" % c), "w") as f:
            f.write("ID1\tID2\tEtc.\n")
            for p in comp["paths"]:
                f.write("\t".join(str(x) for x in p) + "\n")
        with open(os.path.join(out_dir, "edges%d.txt" % c), "w") as f:
            f.write("InID\tOutID\tWeight\tCopycount\tNormalization\n")
            for a, b, w, cc, nm in comp["edges"]:
                f.write("%s\t%s\t%s\t%s\t%s\n" % (a, b, w, cc, nm))
