// Host-side (C++) contig bookkeeping of extension_correction.run_correction, rows a5-a6:
// duplicate_check (extension_correction.py:247-270) and the contig graph by shared K-mers
// (:372-397), sequential over the candidate contigs in seed order exactly as the reference.
// This is control logic over contig bases (the k-mer-level walks are on the GPU, extend.hip); it is
// native code because at 10M reads ~2,000 candidate contigs x ~2,000 bases of Python dict work
// were a third of the host time.
#include "contig_graph.h"

thread_local std::vector<int32_t> g_best_counts;

extern "C" int shn_cgraph_create(int k1, int r, double f, shn_cgraph** out) {
  if (!out || k1 < 2 || r < 1) return shn_fail(SHN_ERR_ARG, "shn_cgraph_create: bad argument");
  *out = new shn_cgraph(k1, r, f);
  return SHN_OK;
}
extern "C" void shn_cgraph_destroy(shn_cgraph* g) { delete g; }
extern "C" int shn_cgraph_add(shn_cgraph* g, const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int32_t* accepted_out, int32_t* best_counts_out) {
  if (!g || (n_cand && (!bases || !off || !accepted_out))) return shn_fail(SHN_ERR_ARG, "shn_cgraph_add: NULL argument");
  std::vector<int32_t> tmp;
  if (!best_counts_out) { tmp.resize(n_cand + 1); best_counts_out = tmp.data(); }
  g->add(bases, off, n_cand, accepted_out, best_counts_out);
  return SHN_OK;
}
extern "C" int shn_cgraph_sizes(const shn_cgraph* g, uint64_t* n_acc, uint64_t* n_conn) {
  if (!g || !n_acc || !n_conn) return shn_fail(SHN_ERR_ARG, "shn_cgraph_sizes: NULL argument");
  uint64_t total = 0;
  for (size_t i = 1; i < g->conns.size(); i++) total += g->conns[i].nb.size();
  *n_acc = g->conns.size() - 1;
  *n_conn = total;
  return SHN_OK;
}
// connections as CSR in *insertion order* (the order Python's dict would iterate): conn_off[n_acc+1], conn_nb[], conn_w[]
extern "C" int shn_cgraph_export(const shn_cgraph* g, uint64_t* conn_off, int32_t* conn_nb, int32_t* conn_w) {
  if (!g || !conn_off || !conn_nb || !conn_w) return shn_fail(SHN_ERR_ARG, "shn_cgraph_export: NULL argument");
  uint64_t p = 0;
  const uint64_t n_acc = g->conns.size() - 1;
  for (uint64_t i = 1; i <= n_acc; i++) {
    conn_off[i - 1] = p;
    for (size_t j = 0; j < g->conns[i].nb.size(); j++) { conn_nb[p] = g->conns[i].nb[j]; conn_w[p] = g->conns[i].w[j]; p++; }
  }
  conn_off[n_acc] = p;
  if (getenv("SHN_DEBUG"))
    fprintf(stderr, "[contig_graph] %llu candidates: %llu evaluations in %llu batches, %.3f s, inserts %.3f s\n",
            (unsigned long long)g->n_cand_total, (unsigned long long)g->n_evals, (unsigned long long)g->n_batches, g->t_eval, g->t_accept);
  return SHN_OK;
}

// contigs: n_cand candidate strings (bases[off[i]..off[i+1])), in seed order, in one call.
// accepted_out[i] = 1-based accepted index or 0.  Connections are returned as CSR in *insertion order*
// (caller sizes conn_* with the value returned in *n_conn after a first call with conn_nb == NULL).
extern "C" int shn_contig_graph(const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                                int32_t* accepted_out, uint64_t* n_acc_out, uint64_t* conn_off, int32_t* conn_nb, int32_t* conn_w,
                                uint64_t* n_conn) {
  if (!bases || !off || !accepted_out || !n_acc_out || !n_conn) return shn_fail(SHN_ERR_ARG, "shn_contig_graph: NULL argument");
  // stateless: the sizing call (conn_nb == NULL) and the fill call each run the stage; callers that want it once use the
  // handle API (shn_cgraph_create / _add / _sizes / _export)
  shn_cgraph g(k1, r, f);
  std::vector<int32_t> bc(n_cand + 1, 0);
  g.add(bases, off, n_cand, accepted_out, bc.data());
  g_best_counts.assign(bc.begin(), bc.begin() + n_cand);
  int rc = shn_cgraph_sizes(&g, n_acc_out, n_conn);
  if (rc) return rc;
  if (conn_nb && conn_w && conn_off) rc = shn_cgraph_export(&g, conn_off, conn_nb, conn_w);
  return rc;
}

// hit count of the `best` contig (max_till_now, extension_correction.py:255-259) of every candidate of this thread's
// last shn_contig_graph call: what the component-sharded duplicate check needs to rule out interference from other shards
extern "C" int shn_contig_best_counts(int32_t* out, uint64_t n_cand) {
  if (!out) return shn_fail(SHN_ERR_ARG, "shn_contig_best_counts: NULL argument");
  if (g_best_counts.size() != n_cand) return shn_fail(SHN_ERR_ARG, "shn_contig_best_counts: no matching shn_contig_graph call");
  memcpy(out, g_best_counts.data(), n_cand * sizeof(int32_t));
  return SHN_OK;
}

// Connected components of the contig graph by the reference's depth-first search (extension_correction.py:417-434): contigs
// 1..n_acc in index order start a component when not yet seen; a component's members come in the order the DFS pops them
// (neighbours pushed in dict insertion order).  comp_of[a-1] = root contig (1-based) of contig a; members[] = the contigs of
// the components one after the other, component j at members[comp_off[j]..comp_off[j+1]); comp_edges[j] = number of distinct
// undirected edges of component j (the METIS header's E, :446-456).  comp_off / comp_edges hold up to n_acc+1 / n_acc entries.
extern "C" int shn_contig_components(uint64_t n_acc, const uint64_t* conn_off, const int32_t* conn_nb, int32_t* comp_of, int32_t* members,
                                     uint64_t* comp_off, uint64_t* comp_edges, uint64_t* n_comp_out) {
  if (!n_comp_out || (n_acc && (!conn_off || !comp_of || !members || !comp_off || !comp_edges)))
    return shn_fail(SHN_ERR_ARG, "shn_contig_components: NULL argument");
  std::vector<uint8_t> seen(n_acc + 1, 0);
  std::vector<int32_t> stack;
  uint64_t n_comp = 0, at = 0;
  if (n_acc) comp_off[0] = 0;
  for (uint64_t ci = 1; ci <= n_acc; ci++) {
    if (seen[ci]) continue;
    stack.assign(1, (int32_t)ci);
    seen[ci] = 1;
    uint64_t deg = 0;
    while (!stack.empty()) {
      const int32_t cur = stack.back();
      stack.pop_back();
      comp_of[cur - 1] = (int32_t)ci;
      members[at++] = cur;
      deg += conn_off[cur] - conn_off[cur - 1];
      for (uint64_t q = conn_off[cur - 1]; q < conn_off[cur]; q++) {
        const int32_t nb = conn_nb[q];
        if (nb < 1 || (uint64_t)nb > n_acc) return shn_fail(SHN_ERR_ARG, "shn_contig_components: neighbour out of range");
        if (!seen[nb]) { stack.push_back(nb); seen[nb] = 1; }
      }
    }
    comp_edges[n_comp] = deg / 2;
    comp_off[++n_comp] = at;
  }
  *n_comp_out = n_comp;
  return SHN_OK;
}
