// Unitig contraction of the raw K-mer graphs of ALL partitions in one batch on the GPU (row a14, first pass):
// replaces load_single_jellyfish (multibridging.py:145-172: a node per K-mer, an edge per k1-mer) followed by the first
// Node.condense_all (mbgraph.py:479-498, Edge.condense :184-257) -- on the host that was 3/4 of the graph stage.
//
// What has to come out identical to the sequential code is not only the set of unitigs but the ORDER of things, which
// the later passes observe: node order (= creation order), and the order of every node's in-/out-edge lists.
//   * nodes: K-mers are numbered in order of first occurrence over the partition's k1-mer rows (contigs in order).
//     condense_all visits `Node.nodes` while it grows: first the K-mers in id order, then the merged nodes in creation
//     order.  A visited node with a condensable out-edge (outdeg(u) = 1, indeg(v) = 1, u != v) is merged with its
//     successor: u and v die, the merged node is appended.  Along a maximal chain of condensable edges this is a
//     process in rounds: round 0 visits the K-mers, round r+1 the nodes created in round r.  In a round, group j of
//     a chain (groups in chain order, pos = visiting order) is alive when visited unless its predecessor was visited
//     earlier AND was alive: a(1) = 1, a(j) = !(pos(j-1) < pos(j) && a(j-1)); an alive group that is not the chain's
//     last merges with what follows it.  Runs of merged links form the next round's groups; a group is created at
//     the time of its last merge (the largest pos among its merging groups).  A chain is finished when one group is
//     left: the final node's creation stamp is (round, time).
//   * edge lists: a merge re-creates the in-edges of its first node and the out-edges of its second node (appended to
//     the lists at their other ends).  So the out-list of a final node is ordered by the time its successors' chains
//     last merged at their HEAD (never: row order, first), the in-list by the time the predecessors' chains were last
//     absorbed at their TAIL; edge ids follow the later of the two.
// Pure cycles of condensable edges (a circular contig) are left to the host code: the partition is flagged.
// Device work: K-mer -> node by a per-partition open-addressing table (first occurrence by atomicMin), degrees by
// atomics, chains by pointer jumping (list ranking), chain-major layout by a scan, then the rounds as scans / segmented
// maxima over the shrinking group array.  Everything is integer work bound by random HBM accesses.
#include "common.h"
#include <functional>
#include <thread>
#include <atomic>
#include "unitigs.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define UG_BLK 256
#define UG_NONE 0xFFFFFFFFu

namespace {

static inline uint32_t ug_grid(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cdiv(n, UG_BLK), 1), 1u << 20); }
#define UG_FOR(i, n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (uint64_t)gridDim.x * blockDim.x)

struct DevBufs {
  std::vector<void*> p;
  template <class T> hipError_t get(T** out, size_t bytes) { hipError_t e = shn_dev_malloc(out, bytes); if (e == hipSuccess) p.push_back((void*)*out); return e; }
  ~DevBufs() { for (void* q : p) shn_dev_free(q); }
};

__device__ __forceinline__ uint32_t ug_code(uint8_t b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; }

__global__ void ug_cid_kernel(const uint64_t* __restrict__ off, uint64_t n_contigs, uint32_t* __restrict__ cid) {
  const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (c >= n_contigs) return;
  for (uint64_t g = off[c] + (threadIdx.x & 63); g < off[c + 1]; g += 64) cid[g] = (uint32_t)c;
}

// K-window at base g: insert its K-mer into the table region of its partition; first occurrence by atomicMin
__global__ void ug_insert_kernel(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, const uint32_t* __restrict__ cid,
                                 const uint32_t* __restrict__ part_of, const uint64_t* __restrict__ tab_off, uint64_t total, int K,
                                 unsigned long long* __restrict__ tkey, uint32_t* __restrict__ tfirst, uint32_t* __restrict__ wslot,
                                 unsigned long long* __restrict__ n_bad) {
  UG_FOR(g, total) {
    const uint32_t c = cid[g];
    const uint64_t end = off[c + 1];
    if (end - off[c] < (uint64_t)K + 1 || g + (uint64_t)K > end) { wslot[g] = UG_NONE; continue; }
    unsigned long long key = 0;
    bool bad = false;
    for (int j = 0; j < K; j++) { const uint32_t b = ug_code(bases[g + j]); bad |= b > 3; key = (key << 2) | (unsigned long long)(b & 3); }
    if (bad) { atomicAdd(n_bad, 1ULL); wslot[g] = UG_NONE; continue; }
    const unsigned long long kk = key + 1ULL;                    // 0 = empty (K <= 31: key < 2^62)
    const uint32_t p = part_of[c];
    const uint64_t lo = tab_off[p], size = tab_off[p + 1] - lo;  // a power of two
    uint64_t s = shn_mix64(key) & (size - 1);
    while (true) {
      unsigned long long cur = __hip_atomic_load(&tkey[lo + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0) { const unsigned long long old = atomicCAS(&tkey[lo + s], 0ULL, kk); cur = old == 0 ? kk : old; }
      if (cur == kk) break;
      s = (s + 1) & (size - 1);
    }
    atomicMin(&tfirst[lo + s], (uint32_t)g);
    wslot[g] = (uint32_t)(lo + s);
  }
}
__global__ void ug_first_flag_kernel(const uint32_t* __restrict__ wslot, const uint32_t* __restrict__ tfirst, uint64_t total, uint32_t* __restrict__ flag) {
  UG_FOR(g, total) flag[g] = (wslot[g] != UG_NONE && tfirst[wslot[g]] == (uint32_t)g) ? 1u : 0u;
}
// node id of a table slot = rank of its first occurrence; nfirst[node] = that window
__global__ void ug_node_ids_kernel(const uint32_t* __restrict__ flag, const uint64_t* __restrict__ rank, const uint32_t* __restrict__ wslot, uint64_t total,
                                   uint32_t* __restrict__ slot_node, uint32_t* __restrict__ nfirst) {
  UG_FOR(g, total) if (flag[g]) { slot_node[wslot[g]] = (uint32_t)rank[g]; nfirst[rank[g]] = (uint32_t)g; }
}
// one edge per k1-window (row): degrees, and the edge itself remembered at both ends (meaningful where the degree is 1)
__global__ void ug_degree_kernel(const uint32_t* __restrict__ wslot, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ cid,
                                 const uint64_t* __restrict__ off, uint64_t total, int K, uint32_t* __restrict__ outdeg, uint32_t* __restrict__ indeg,
                                 uint32_t* __restrict__ out_to, uint32_t* __restrict__ in_from) {
  UG_FOR(g, total) {
    if (wslot[g] == UG_NONE || g + (uint64_t)K + 1 > off[cid[g] + 1]) continue;
    const uint32_t u = slot_node[wslot[g]], v = slot_node[wslot[g + 1]];
    atomicAdd(&outdeg[u], 1u);
    atomicAdd(&indeg[v], 1u);
    out_to[u] = v;
    in_from[v] = u;
  }
}
// condensable links: succ / pred along the chains
__global__ void ug_links_kernel(const uint32_t* __restrict__ outdeg, const uint32_t* __restrict__ indeg, const uint32_t* __restrict__ out_to, uint64_t nn,
                                uint32_t* __restrict__ succ, uint32_t* __restrict__ pred) {
  UG_FOR(u, nn) {
    if (outdeg[u] != 1) continue;
    const uint32_t v = out_to[u];
    if (indeg[v] == 1 && v != (uint32_t)u) { succ[u] = v; pred[v] = (uint32_t)u; }
  }
}
// list ranking by pointer jumping: (jump[u], dist[u]) = an ancestor along pred and the distance to it
__global__ void ug_jump_init_kernel(const uint32_t* __restrict__ pred, uint64_t nn, uint32_t* __restrict__ jump, uint32_t* __restrict__ dist) {
  UG_FOR(u, nn) { const uint32_t p = pred[u]; jump[u] = p == UG_NONE ? (uint32_t)u : p; dist[u] = p == UG_NONE ? 0u : 1u; }
}
__global__ void ug_jump_kernel(const uint32_t* __restrict__ jin, const uint32_t* __restrict__ din, uint64_t nn, uint32_t* __restrict__ jout, uint32_t* __restrict__ dout) {
  UG_FOR(u, nn) { const uint32_t j = jin[u]; jout[u] = jin[j]; dout[u] = din[u] + din[j]; }
}
// after the jumps: jump[u] is the chain's head unless u lies on a cycle; chain length by atomicMax at the head
__global__ void ug_chain_len_kernel(const uint32_t* __restrict__ jump, const uint32_t* __restrict__ dist, const uint32_t* __restrict__ pred,
                                    const uint32_t* __restrict__ part_of_node, uint64_t nn, uint32_t* __restrict__ clen, uint32_t* __restrict__ cyclic) {
  UG_FOR(u, nn) {
    const uint32_t h = jump[u];
    if (pred[h] != UG_NONE) { cyclic[part_of_node[u]] = 1; continue; }
    atomicMax(&clen[h], dist[u] + 1);
  }
}
__global__ void ug_node_part_kernel(const uint32_t* __restrict__ nfirst, const uint32_t* __restrict__ cid, const uint32_t* __restrict__ part_of, uint64_t nn,
                                    uint32_t* __restrict__ part_of_node) {
  UG_FOR(u, nn) part_of_node[u] = part_of[cid[nfirst[u]]];
}
// heads of chains with >= 2 K-mers: flag for the scans (chain index, first group slot)
__global__ void ug_multi_flag_kernel(const uint32_t* __restrict__ clen, uint64_t nn, uint32_t* __restrict__ is_multi, uint32_t* __restrict__ multi_len) {
  UG_FOR(u, nn) { const bool m = clen[u] >= 2; is_multi[u] = m ? 1u : 0u; multi_len[u] = m ? clen[u] : 0u; }
}
// chain-major layout of the K-mers of the multi-K-mer chains: slot = chain_off[head] + dist
__global__ void ug_layout_kernel(const uint32_t* __restrict__ jump, const uint32_t* __restrict__ dist, const uint32_t* __restrict__ clen,
                                 const uint64_t* __restrict__ chain_off, const uint64_t* __restrict__ chain_idx, uint64_t nn,
                                 uint32_t* __restrict__ lay_node, uint32_t* __restrict__ gchain, uint64_t* __restrict__ gpos,
                                 uint64_t* __restrict__ coff, uint32_t* __restrict__ chain_head) {
  UG_FOR(u, nn) {
    const uint32_t h = jump[u];
    if (clen[h] < 2) continue;
    const uint64_t at = chain_off[h] + dist[u];
    lay_node[at] = (uint32_t)u;
    gchain[at] = (uint32_t)chain_idx[h];
    gpos[at] = (uint64_t)u;                         // round 0: the K-mers are visited in id order
    if ((uint32_t)u == h) { coff[chain_idx[h]] = chain_off[h]; chain_head[chain_idx[h]] = h; }
  }
}

// ---- one round over the group array (chain-major; coff[c] .. coff[c+1] are chain c's groups)
// reset[i] = 1 where an ascending run of visiting positions starts (chain start, or predecessor visited later)
__global__ void ug_reset_kernel(const uint32_t* __restrict__ gchain, const uint64_t* __restrict__ gpos, const uint64_t* __restrict__ coff, uint64_t ng,
                                uint32_t* __restrict__ reset) {
  UG_FOR(i, ng) reset[i] = (i == coff[gchain[i]] || !(gpos[i - 1] < gpos[i])) ? 1u : 0u;
}
__global__ void ug_run_start_kernel(const uint32_t* __restrict__ reset, const uint64_t* __restrict__ rsum, uint64_t ng, uint64_t* __restrict__ starts) {
  UG_FOR(i, ng) if (reset[i]) starts[rsum[i]] = i;            // rsum = exclusive scan: the index of this run
}
// merged[i] = group i is alive when visited and not the chain's last: it merges with what follows.
// boundary[i] = a new group starts at i.  Records the head / tail merge times of the chain for the edge lists.
__global__ void ug_alive_kernel(const uint32_t* __restrict__ reset, const uint64_t* __restrict__ rsum, const uint64_t* __restrict__ starts,
                                const uint32_t* __restrict__ gchain, const uint64_t* __restrict__ gpos, const uint64_t* __restrict__ coff, uint64_t ng,
                                uint64_t round_tag, uint32_t* __restrict__ merged, uint64_t* __restrict__ t_head, uint64_t* __restrict__ t_tail) {
  UG_FOR(i, ng) {
    const uint32_t c = gchain[i];
    const uint64_t run = reset[i] ? rsum[i] : rsum[i] - 1;      // exclusive scan counts the resets before i
    const bool alive = ((i - starts[run]) & 1ULL) == 0;
    const bool last = i + 1 == coff[c + 1];
    const bool m = alive && !last;
    merged[i] = m ? 1u : 0u;
    if (i == coff[c]) t_head[c] = round_tag | gpos[i];          // (a chain in the array has >= 2 groups: its head merges)
    if (m && i + 2 == coff[c + 1]) t_tail[c] = round_tag | gpos[i];
  }
}
__global__ void ug_boundary_kernel(const uint32_t* __restrict__ merged, const uint32_t* __restrict__ gchain, const uint64_t* __restrict__ coff, uint64_t ng,
                                   uint32_t* __restrict__ boundary) {
  UG_FOR(i, ng) boundary[i] = (i == coff[gchain[i]] || !merged[i - 1]) ? 1u : 0u;
}
// creation time of the new groups: the largest visiting position among their merging groups
__global__ void ug_newpos_kernel(const uint32_t* __restrict__ merged, const uint32_t* __restrict__ boundary, const uint64_t* __restrict__ bsum,
                                 const uint64_t* __restrict__ gpos, uint64_t ng, unsigned long long* __restrict__ newpos) {
  UG_FOR(i, ng) {
    const uint64_t g = boundary[i] ? bsum[i] : bsum[i] - 1;
    if (merged[i]) atomicMax(&newpos[g], (unsigned long long)gpos[i]);
  }
}
// groups per chain after the round
__global__ void ug_chain_count_kernel(const uint32_t* __restrict__ boundary, const uint32_t* __restrict__ gchain, uint64_t ng, uint32_t* __restrict__ ccount) {
  UG_FOR(i, ng) if (boundary[i]) atomicAdd(&ccount[gchain[i]], 1u);
}
// chains left with one group are finished (creation stamp recorded); the others keep their groups for the next round
__global__ void ug_chain_done_kernel(const uint32_t* __restrict__ ccount, const uint32_t* __restrict__ chain_head_in, const uint64_t* __restrict__ t_head_in,
                                     const uint64_t* __restrict__ t_tail_in, const uint64_t* __restrict__ coff, const uint32_t* __restrict__ boundary,
                                     const uint64_t* __restrict__ bsum, const unsigned long long* __restrict__ newpos, uint64_t nc, uint64_t round_tag,
                                     uint64_t* __restrict__ fin_stamp, uint64_t* __restrict__ fin_thead, uint64_t* __restrict__ fin_ttail,
                                     uint32_t* __restrict__ keep, uint32_t* __restrict__ keep_groups) {
  UG_FOR(c, nc) {
    const bool done = ccount[c] == 1;
    keep[c] = done ? 0u : 1u;
    keep_groups[c] = done ? 0u : ccount[c];
    if (done) {
      const uint32_t h = chain_head_in[c];
      const uint64_t g = bsum[coff[c]];                            // (coff[c] is a boundary: exclusive scan = its group index)
      fin_stamp[h] = round_tag | (uint64_t)newpos[g];
      fin_thead[h] = t_head_in[c];
      fin_ttail[h] = t_tail_in[c];
    }
  }
}
// carry the unfinished chains over: new chain index, new group slots
__global__ void ug_carry_chains_kernel(const uint32_t* __restrict__ keep, const uint64_t* __restrict__ ksum, const uint64_t* __restrict__ gsum,
                                       const uint32_t* __restrict__ chain_head_in, const uint64_t* __restrict__ t_head_in, const uint64_t* __restrict__ t_tail_in,
                                       uint64_t nc, uint32_t* __restrict__ chain_head_out, uint64_t* __restrict__ t_head_out, uint64_t* __restrict__ t_tail_out,
                                       uint64_t* __restrict__ coff_out) {
  UG_FOR(c, nc) if (keep[c]) {
    const uint64_t n = ksum[c];
    chain_head_out[n] = chain_head_in[c]; t_head_out[n] = t_head_in[c]; t_tail_out[n] = t_tail_in[c];
    coff_out[n] = gsum[c];
  }
}
__global__ void ug_carry_groups_kernel(const uint32_t* __restrict__ boundary, const uint64_t* __restrict__ bsum, const uint32_t* __restrict__ gchain,
                                       const uint32_t* __restrict__ keep, const uint64_t* __restrict__ ksum, const uint64_t* __restrict__ gsum,
                                       const uint64_t* __restrict__ coff, const unsigned long long* __restrict__ newpos, const uint64_t* __restrict__ gpos,
                                       const uint32_t* __restrict__ merged, uint64_t ng, uint32_t* __restrict__ gchain_out, uint64_t* __restrict__ gpos_out) {
  UG_FOR(i, ng) {
    if (!boundary[i]) continue;
    const uint32_t c = gchain[i];
    if (!keep[c]) continue;
    const uint64_t g = bsum[i];
    const uint64_t at = gsum[c] + (g - bsum[coff[c]]);
    gchain_out[at] = (uint32_t)ksum[c];
    // a group that merged nothing this round (the chain's last group, not absorbed) keeps its old position; it is never
    // a merging group again before it is absorbed, so the value is not compared with this round's creation times
    gpos_out[at] = merged[i] ? (uint64_t)newpos[g] : gpos[i];
  }
}

// ---- products
// per K-mer node: chain head (itself for a single K-mer); per head: final-node record
__global__ void ug_fin_single_kernel(const uint32_t* __restrict__ jump, const uint32_t* __restrict__ clen, uint64_t nn, uint64_t* __restrict__ fin_stamp) {
  UG_FOR(u, nn) if (jump[u] == (uint32_t)u && clen[u] == 1) fin_stamp[u] = (uint64_t)u;    // round tag 0: before every merged node
}
__global__ void ug_head_flag_kernel(const uint32_t* __restrict__ jump, const uint32_t* __restrict__ pred, uint64_t nn, uint32_t* __restrict__ flag,
                                    uint32_t* __restrict__ blen, const uint32_t* __restrict__ clen, int K) {
  UG_FOR(u, nn) { const bool h = jump[u] == (uint32_t)u && pred[u] == UG_NONE; flag[u] = h ? 1u : 0u; blen[u] = h ? (uint32_t)K + clen[u] - 1 : 0u; }
}
// bases of the final nodes: the head's K-mer, then the last base of every further K-mer of the chain
__global__ void ug_bases_kernel(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ nfirst, const uint32_t* __restrict__ jump,
                                const uint32_t* __restrict__ dist, const uint32_t* __restrict__ pred, const uint64_t* __restrict__ boff, uint64_t nn, int K,
                                uint8_t* __restrict__ out) {
  UG_FOR(u, nn) {
    const uint32_t h = jump[u];
    if (pred[h] != UG_NONE) continue;                               // on a cycle
    const uint64_t o = boff[h];
    const uint64_t g = nfirst[u];
    if ((uint32_t)u == h) for (int j = 0; j < K; j++) out[o + j] = bases[g + j];
    else out[o + K - 1 + dist[u]] = bases[g + K - 1];
  }
}
// edges between final nodes: the rows that are not condensable links
__global__ void ug_edge_flag_kernel(const uint32_t* __restrict__ wslot, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ cid,
                                    const uint64_t* __restrict__ off, const uint32_t* __restrict__ succ, uint64_t total, int K, uint32_t* __restrict__ flag) {
  UG_FOR(g, total) {
    uint32_t f = 0;
    if (wslot[g] != UG_NONE && g + (uint64_t)K + 1 <= off[cid[g] + 1]) {
      const uint32_t u = slot_node[wslot[g]], v = slot_node[wslot[g + 1]];
      f = succ[u] == v ? 0u : 1u;
    }
    flag[g] = f;
  }
}
__global__ void ug_edges_kernel(const uint32_t* __restrict__ flag, const uint64_t* __restrict__ epos, const uint32_t* __restrict__ wslot,
                                const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ jump, uint64_t total, uint32_t* __restrict__ e_src,
                                uint32_t* __restrict__ e_dst) {
  UG_FOR(g, total) if (flag[g]) {
    const uint32_t u = slot_node[wslot[g]], v = slot_node[wslot[g + 1]];
    e_src[epos[g]] = jump[u];                   // chain heads (u is its chain's last K-mer, v its chain's first)
    e_dst[epos[g]] = jump[v];
  }
}

// out-degree of a chain's last K-mer, recorded at its head
__global__ void ug_tail_out_kernel(const uint32_t* __restrict__ jump, const uint32_t* __restrict__ dist, const uint32_t* __restrict__ clen,
                                   const uint32_t* __restrict__ pred, const uint32_t* __restrict__ outdeg, uint64_t nn, uint32_t* __restrict__ tail_out) {
  UG_FOR(u, nn) { const uint32_t h = jump[u]; if (pred[h] == UG_NONE && dist[u] + 1 == clen[h]) tail_out[h] = outdeg[u]; }
}
// the final nodes (chain heads) compacted in K-mer id order = partition by partition
__global__ void ug_head_compact_kernel(const uint32_t* __restrict__ hflag, const uint64_t* __restrict__ hpos, const uint64_t* __restrict__ stamp,
                                       const uint64_t* __restrict__ thead, const uint64_t* __restrict__ ttail, const uint32_t* __restrict__ clen,
                                       const uint32_t* __restrict__ tail_out, const uint32_t* __restrict__ part_of_node, const uint64_t* __restrict__ boff,
                                       uint64_t nn, uint64_t* __restrict__ o_stamp, uint64_t* __restrict__ o_thead, uint64_t* __restrict__ o_ttail,
                                       uint32_t* __restrict__ o_clen, uint32_t* __restrict__ o_tail_out, uint32_t* __restrict__ o_part, uint64_t* __restrict__ o_boff) {
  UG_FOR(u, nn) if (hflag[u]) {
    const uint64_t i = hpos[u];
    o_stamp[i] = stamp[u]; o_thead[i] = thead[u]; o_ttail[i] = ttail[u]; o_clen[i] = clen[u]; o_tail_out[i] = tail_out[u];
    o_part[i] = part_of_node[u]; o_boff[i] = boff[u];
  }
}
// edges as (compact head index of the source chain, of the destination chain)
__global__ void ug_edge_heads_kernel(uint32_t* __restrict__ e_src, uint32_t* __restrict__ e_dst, const uint64_t* __restrict__ hpos, uint64_t ne) {
  UG_FOR(x, ne) { e_src[x] = (uint32_t)hpos[e_src[x]]; e_dst[x] = (uint32_t)hpos[e_dst[x]]; }
}

}  // namespace

extern "C" void shn_unitigs_destroy(shn_unitigs* u) { delete u; }
extern "C" uint64_t shn_unitigs_n_kmers(const shn_unitigs* u, uint32_t part) { return (u && part < u->n_parts) ? u->n_kmers[part] : 0; }

// bases/off: the contigs of all partitions one after the other (ASCII), part_of[c] = partition of contig c, ascending.
extern "C" int shn_unitigs_build(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_contigs, const uint32_t* part_of, uint32_t n_parts,
                                 int K, shn_unitigs** out) {
  if (!ctx || !out || (n_contigs && (!bases || !off || !part_of)) || K < 2 || K > 31) return shn_fail(SHN_ERR_ARG, "shn_unitigs_build: bad argument");
  *out = nullptr;
  const bool dbg = getenv("SHN_DEBUG") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char* what) { if (dbg) { double t = now(); fprintf(stderr, "[unitigs] %-34s %8.3f s\n", what, t - t0); t0 = t; } };
  shn_unitigs* U = new shn_unitigs();
  struct Guard { shn_unitigs* u; ~Guard() { delete u; } } guard{U};
  U->K = K; U->n_parts = n_parts;
  U->n_kmers.assign(n_parts, 0); U->cyclic.assign(n_parts, 0);
  U->node_off.assign(n_parts + 1, 0); U->edge_off.assign(n_parts + 1, 0); U->base_off.assign(1, 0);
  const uint64_t total = n_contigs ? off[n_contigs] : 0;
  if (!total) { *out = U; guard.u = nullptr; return SHN_OK; }
  if (off[0] != 0) return shn_fail(SHN_ERR_ARG, "shn_unitigs_build: offsets must start at 0");
  if (total >= 0x7FFFFFF0ULL) return shn_fail(SHN_ERR_OVERFLOW, "shn_unitigs_build: more than 2^31 contig bases");
  for (uint64_t c = 0; c < n_contigs; c++)
    if (part_of[c] >= n_parts || (c && part_of[c] < part_of[c - 1])) return shn_fail(SHN_ERR_ARG, "shn_unitigs_build: part_of must be ascending and < n_parts");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  shn_stage_begin(ctx);
  TimerRegion treg(ctx, T_GRAPH_GPU);
  // table regions: a power of two >= 2 x the K-windows of the partition
  std::vector<uint64_t> tab_off(n_parts + 1, 0), win(n_parts, 0), first_base(n_parts + 1, total);
  for (uint64_t c = 0; c < n_contigs; c++) {
    const uint64_t L = off[c + 1] - off[c];
    if (L >= (uint64_t)K + 1) win[part_of[c]] += L - K + 1;
    first_base[part_of[c]] = std::min(first_base[part_of[c]], off[c]);
  }
  for (int64_t p = (int64_t)n_parts - 1; p >= 0; p--) if (first_base[p] == total) first_base[p] = first_base[p + 1];      // (empty partitions)
  for (uint32_t p = 0; p < n_parts; p++) { uint64_t sz = 16; while (sz < 2 * win[p]) sz <<= 1; tab_off[p + 1] = tab_off[p] + sz; }
  const uint64_t slots = tab_off[n_parts];
  if (slots >= 0xFFFFFFF0ULL) return shn_fail(SHN_ERR_OVERFLOW, "shn_unitigs_build: K-mer table beyond 2^32 slots");
  DevBufs B;
  uint8_t* d_bases; uint64_t *d_off, *d_tab_off; uint32_t *d_cid, *d_part_of, *d_wslot, *d_tfirst, *d_flag; unsigned long long *d_tkey, *d_bad; uint64_t* d_scan;
  HIP_TRY(B.get(&d_bases, total + 64)); HIP_TRY(B.get(&d_off, (n_contigs + 1) * 8)); HIP_TRY(B.get(&d_tab_off, (n_parts + 1) * 8));
  HIP_TRY(B.get(&d_cid, (total + 1) * 4)); HIP_TRY(B.get(&d_part_of, n_contigs * 4)); HIP_TRY(B.get(&d_wslot, (total + 2) * 4));
  HIP_TRY(B.get(&d_tkey, slots * 8)); HIP_TRY(B.get(&d_tfirst, slots * 4)); HIP_TRY(B.get(&d_flag, (total + 1) * 4)); HIP_TRY(B.get(&d_bad, 8));
  HIP_TRY(B.get(&d_scan, (total + 2) * 8));
  HIP_TRY(hipMemcpyAsync(d_bases, bases, total, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_off, off, (n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_tab_off, tab_off.data(), (n_parts + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_part_of, part_of, n_contigs * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_tkey, 0, slots * 8, s));
  HIP_TRY(hipMemsetAsync(d_tfirst, 0xFF, slots * 4, s));
  HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
  HIP_TRY(hipMemsetAsync(d_wslot + total, 0xFF, 8, s));
  hipLaunchKernelGGL(ug_cid_kernel, dim3((uint32_t)cdiv(n_contigs * 64, UG_BLK)), dim3(UG_BLK), 0, s, d_off, n_contigs, d_cid);
  hipLaunchKernelGGL(ug_insert_kernel, dim3(ug_grid(total)), dim3(UG_BLK), 0, s, d_bases, d_off, d_cid, d_part_of, d_tab_off, total, K, d_tkey, d_tfirst, d_wslot, d_bad);
  hipLaunchKernelGGL(ug_first_flag_kernel, dim3(ug_grid(total)), dim3(UG_BLK), 0, s, d_wslot, d_tfirst, total, d_flag);
  uint64_t nn = 0;
  int rc = shn_device_scan_u32(ctx, d_flag, total, d_scan, &nn);
  if (rc) return rc;
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
  if (bad) return shn_fail(SHN_ERR_ARG, "shn_unitigs_build: contig with a base outside ACGT");
  // K-mer nodes per partition: the rank at the partition's first base
  {
    std::vector<uint64_t> at(n_parts + 1, nn);
    for (uint32_t p = 0; p < n_parts; p++) if (first_base[p] < total) HIP_TRY(hipMemcpyAsync(&at[p], d_scan + first_base[p], 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int64_t p = (int64_t)n_parts - 1; p >= 0; p--) if (first_base[p] >= total) at[p] = at[p + 1];
    for (uint32_t p = 0; p < n_parts; p++) U->n_kmers[p] = at[p + 1] - at[p];
  }
  if (!nn) { *out = U; guard.u = nullptr; return SHN_OK; }
  lap("K-mer table, node ids");
  uint32_t *d_slot_node, *d_nfirst, *d_outdeg, *d_indeg, *d_out_to, *d_in_from, *d_succ, *d_pred, *d_jump[2], *d_dist[2], *d_clen, *d_node_part, *d_cyc;
  HIP_TRY(B.get(&d_slot_node, slots * 4)); HIP_TRY(B.get(&d_nfirst, nn * 4)); HIP_TRY(B.get(&d_outdeg, nn * 4)); HIP_TRY(B.get(&d_indeg, nn * 4));
  HIP_TRY(B.get(&d_out_to, nn * 4)); HIP_TRY(B.get(&d_in_from, nn * 4)); HIP_TRY(B.get(&d_succ, nn * 4)); HIP_TRY(B.get(&d_pred, nn * 4));
  for (int i = 0; i < 2; i++) { HIP_TRY(B.get(&d_jump[i], nn * 4)); HIP_TRY(B.get(&d_dist[i], nn * 4)); }
  HIP_TRY(B.get(&d_clen, nn * 4)); HIP_TRY(B.get(&d_node_part, nn * 4)); HIP_TRY(B.get(&d_cyc, (n_parts + 1) * 4));
  HIP_TRY(hipMemsetAsync(d_outdeg, 0, nn * 4, s)); HIP_TRY(hipMemsetAsync(d_indeg, 0, nn * 4, s));
  HIP_TRY(hipMemsetAsync(d_succ, 0xFF, nn * 4, s)); HIP_TRY(hipMemsetAsync(d_pred, 0xFF, nn * 4, s));
  HIP_TRY(hipMemsetAsync(d_clen, 0, nn * 4, s)); HIP_TRY(hipMemsetAsync(d_cyc, 0, (n_parts + 1) * 4, s));
  hipLaunchKernelGGL(ug_node_ids_kernel, dim3(ug_grid(total)), dim3(UG_BLK), 0, s, d_flag, d_scan, d_wslot, total, d_slot_node, d_nfirst);
  hipLaunchKernelGGL(ug_node_part_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_nfirst, d_cid, d_part_of, nn, d_node_part);
  hipLaunchKernelGGL(ug_degree_kernel, dim3(ug_grid(total)), dim3(UG_BLK), 0, s, d_wslot, d_slot_node, d_cid, d_off, total, K, d_outdeg, d_indeg, d_out_to, d_in_from);
  hipLaunchKernelGGL(ug_links_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_outdeg, d_indeg, d_out_to, nn, d_succ, d_pred);
  hipLaunchKernelGGL(ug_jump_init_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_pred, nn, d_jump[0], d_dist[0]);
  int cur = 0;
  for (uint64_t span = 1; span < nn; span <<= 1) {
    hipLaunchKernelGGL(ug_jump_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_jump[cur], d_dist[cur], nn, d_jump[cur ^ 1], d_dist[cur ^ 1]);
    cur ^= 1;
    uint64_t max_contig = 0;
    (void)max_contig;
    if (span >= (1ULL << 24)) break;               // (a chain is no longer than its partition's contigs; 2^24 jumps of slack)
  }
  uint32_t *jump = d_jump[cur], *dist = d_dist[cur];
  hipLaunchKernelGGL(ug_chain_len_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, jump, dist, d_pred, d_node_part, nn, d_clen, d_cyc);
  {
    std::vector<uint32_t> cyc(n_parts);
    HIP_TRY(hipMemcpyAsync(cyc.data(), d_cyc, n_parts * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t p = 0; p < n_parts; p++) U->cyclic[p] = cyc[p] ? 1 : 0;
  }
  lap("degrees, chains (list ranking)");
  // ---- the rounds over the multi-K-mer chains
  uint64_t *d_fin_stamp, *d_fin_thead, *d_fin_ttail;
  HIP_TRY(B.get(&d_fin_stamp, nn * 8)); HIP_TRY(B.get(&d_fin_thead, nn * 8)); HIP_TRY(B.get(&d_fin_ttail, nn * 8));
  HIP_TRY(hipMemsetAsync(d_fin_stamp, 0, nn * 8, s)); HIP_TRY(hipMemsetAsync(d_fin_thead, 0, nn * 8, s)); HIP_TRY(hipMemsetAsync(d_fin_ttail, 0, nn * 8, s));
  hipLaunchKernelGGL(ug_fin_single_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, jump, d_clen, nn, d_fin_stamp);
  uint64_t n_rounds = 0;
  {
    DevBufs R;
    uint32_t *d_is_multi, *d_multi_len;
    uint64_t *d_cidx, *d_coff0;
    HIP_TRY(R.get(&d_is_multi, (nn + 1) * 4)); HIP_TRY(R.get(&d_multi_len, (nn + 1) * 4)); HIP_TRY(R.get(&d_cidx, (nn + 2) * 8)); HIP_TRY(R.get(&d_coff0, (nn + 2) * 8));
    hipLaunchKernelGGL(ug_multi_flag_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_clen, nn, d_is_multi, d_multi_len);
    uint64_t nc = 0, ng = 0;
    if ((rc = shn_device_scan_u32(ctx, d_is_multi, nn, d_cidx, &nc)) || (rc = shn_device_scan_u32(ctx, d_multi_len, nn, d_coff0, &ng))) return rc;
    if (nc) {
      // ping-pong group / chain arrays
      uint32_t *gchain[2], *chead[2], *d_lay, *d_reset, *d_merged, *d_boundary, *d_ccount, *d_keep, *d_keepg;
      uint64_t *gpos[2], *coff[2], *thead[2], *ttail[2], *d_rsum, *d_starts, *d_bsum, *d_ksum, *d_gsum;
      unsigned long long* d_newpos;
      for (int i = 0; i < 2; i++) {
        HIP_TRY(R.get(&gchain[i], (ng + 1) * 4)); HIP_TRY(R.get(&gpos[i], (ng + 1) * 8)); HIP_TRY(R.get(&coff[i], (nc + 2) * 8));
        HIP_TRY(R.get(&chead[i], (nc + 1) * 4)); HIP_TRY(R.get(&thead[i], (nc + 1) * 8)); HIP_TRY(R.get(&ttail[i], (nc + 1) * 8));
      }
      HIP_TRY(R.get(&d_lay, (ng + 1) * 4)); HIP_TRY(R.get(&d_reset, (ng + 1) * 4)); HIP_TRY(R.get(&d_merged, (ng + 1) * 4)); HIP_TRY(R.get(&d_boundary, (ng + 1) * 4));
      HIP_TRY(R.get(&d_ccount, (nc + 1) * 4)); HIP_TRY(R.get(&d_keep, (nc + 1) * 4)); HIP_TRY(R.get(&d_keepg, (nc + 1) * 4));
      HIP_TRY(R.get(&d_rsum, (ng + 2) * 8)); HIP_TRY(R.get(&d_starts, (ng + 2) * 8)); HIP_TRY(R.get(&d_bsum, (ng + 2) * 8));
      HIP_TRY(R.get(&d_ksum, (nc + 2) * 8)); HIP_TRY(R.get(&d_gsum, (nc + 2) * 8)); HIP_TRY(R.get(&d_newpos, (ng + 1) * 8));
      HIP_TRY(hipMemsetAsync(thead[0], 0, (nc + 1) * 8, s)); HIP_TRY(hipMemsetAsync(ttail[0], 0, (nc + 1) * 8, s));
      hipLaunchKernelGGL(ug_layout_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, jump, dist, d_clen, d_coff0, d_cidx, nn, d_lay, gchain[0], gpos[0], coff[0], chead[0]);
      HIP_TRY(hipMemcpyAsync(coff[0] + nc, &ng, 8, hipMemcpyHostToDevice, s));
      int a = 0;
      while (nc) {
        n_rounds++;
        if (n_rounds > 4096) return shn_fail(SHN_ERR_INTERNAL, "shn_unitigs_build: the merge rounds do not end");
        const uint64_t tag = n_rounds << 40;                       // stamps: round in the high bits (round 0 = single K-mers)
        uint64_t nruns = 0, ngn = 0, ncn = 0, ngk = 0;
        hipLaunchKernelGGL(ug_reset_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, gchain[a], gpos[a], coff[a], ng, d_reset);
        if ((rc = shn_device_scan_u32(ctx, d_reset, ng, d_rsum, &nruns))) return rc;
        hipLaunchKernelGGL(ug_run_start_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, d_reset, d_rsum, ng, d_starts);
        hipLaunchKernelGGL(ug_alive_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, d_reset, d_rsum, d_starts, gchain[a], gpos[a], coff[a], ng, tag, d_merged,
                           thead[a], ttail[a]);
        hipLaunchKernelGGL(ug_boundary_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, d_merged, gchain[a], coff[a], ng, d_boundary);
        if ((rc = shn_device_scan_u32(ctx, d_boundary, ng, d_bsum, &ngn))) return rc;
        HIP_TRY(hipMemsetAsync(d_newpos, 0, (ngn + 1) * 8, s));
        HIP_TRY(hipMemsetAsync(d_ccount, 0, (nc + 1) * 4, s));
        hipLaunchKernelGGL(ug_newpos_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, d_merged, d_boundary, d_bsum, gpos[a], ng, d_newpos);
        hipLaunchKernelGGL(ug_chain_count_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, d_boundary, gchain[a], ng, d_ccount);
        hipLaunchKernelGGL(ug_chain_done_kernel, dim3(ug_grid(nc)), dim3(UG_BLK), 0, s, d_ccount, chead[a], thead[a], ttail[a], coff[a], d_boundary, d_bsum, d_newpos,
                           nc, tag, d_fin_stamp, d_fin_thead, d_fin_ttail, d_keep, d_keepg);
        if ((rc = shn_device_scan_u32(ctx, d_keep, nc, d_ksum, &ncn)) || (rc = shn_device_scan_u32(ctx, d_keepg, nc, d_gsum, &ngk))) return rc;
        if (ncn) {
          hipLaunchKernelGGL(ug_carry_chains_kernel, dim3(ug_grid(nc)), dim3(UG_BLK), 0, s, d_keep, d_ksum, d_gsum, chead[a], thead[a], ttail[a], nc, chead[a ^ 1],
                             thead[a ^ 1], ttail[a ^ 1], coff[a ^ 1]);
          HIP_TRY(hipMemcpyAsync(coff[a ^ 1] + ncn, &ngk, 8, hipMemcpyHostToDevice, s));
          hipLaunchKernelGGL(ug_carry_groups_kernel, dim3(ug_grid(ng)), dim3(UG_BLK), 0, s, d_boundary, d_bsum, gchain[a], d_keep, d_ksum, d_gsum, coff[a], d_newpos,
                             gpos[a], d_merged, ng, gchain[a ^ 1], gpos[a ^ 1]);
          HIP_TRY(hipStreamSynchronize(s));          // (ngk lives on the host stack)
        }
        nc = ncn; ng = ngk; a ^= 1;
      }
    }
  }
  if (dbg) fprintf(stderr, "[unitigs] %llu K-mers in %u partitions: %llu merge rounds\n", (unsigned long long)nn, n_parts, (unsigned long long)n_rounds);
  lap("merge rounds");
  // ---- products: final nodes (chain heads), their bases, the edges between them
  uint32_t *d_hflag, *d_blen, *d_eflag; uint64_t *d_hpos, *d_boff, *d_epos; uint8_t* d_obases;
  HIP_TRY(B.get(&d_hflag, (nn + 1) * 4)); HIP_TRY(B.get(&d_blen, (nn + 1) * 4)); HIP_TRY(B.get(&d_hpos, (nn + 2) * 8)); HIP_TRY(B.get(&d_boff, (nn + 2) * 8));
  hipLaunchKernelGGL(ug_head_flag_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, jump, d_pred, nn, d_hflag, d_blen, d_clen, K);
  uint64_t nf = 0, nb = 0, ne = 0;
  if ((rc = shn_device_scan_u32(ctx, d_hflag, nn, d_hpos, &nf)) || (rc = shn_device_scan_u32(ctx, d_blen, nn, d_boff, &nb))) return rc;
  HIP_TRY(B.get(&d_obases, nb + 64));
  hipLaunchKernelGGL(ug_bases_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_bases, d_nfirst, jump, dist, d_pred, d_boff, nn, K, d_obases);
  HIP_TRY(B.get(&d_eflag, (total + 1) * 4)); HIP_TRY(B.get(&d_epos, (total + 2) * 8));
  hipLaunchKernelGGL(ug_edge_flag_kernel, dim3(ug_grid(total)), dim3(UG_BLK), 0, s, d_wslot, d_slot_node, d_cid, d_off, d_succ, total, K, d_eflag);
  if ((rc = shn_device_scan_u32(ctx, d_eflag, total, d_epos, &ne))) return rc;
  uint32_t *d_esrc = nullptr, *d_edst = nullptr;
  if (ne) {
    HIP_TRY(B.get(&d_esrc, ne * 4)); HIP_TRY(B.get(&d_edst, ne * 4));
    hipLaunchKernelGGL(ug_edges_kernel, dim3(ug_grid(total)), dim3(UG_BLK), 0, s, d_eflag, d_epos, d_wslot, d_slot_node, jump, total, d_esrc, d_edst);
  }
  // the final nodes' records, compacted on the device (chain heads in K-mer id order, i.e. partition by partition)
  uint32_t *d_tail_out, *d_oclen, *d_otail, *d_opart; uint64_t *d_ostamp, *d_othead, *d_ottail, *d_oboff;
  HIP_TRY(B.get(&d_tail_out, nn * 4));
  HIP_TRY(hipMemsetAsync(d_tail_out, 0, nn * 4, s));
  hipLaunchKernelGGL(ug_tail_out_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, jump, dist, d_clen, d_pred, d_outdeg, nn, d_tail_out);
  HIP_TRY(B.get(&d_oclen, (nf + 1) * 4)); HIP_TRY(B.get(&d_otail, (nf + 1) * 4)); HIP_TRY(B.get(&d_opart, (nf + 1) * 4));
  HIP_TRY(B.get(&d_ostamp, (nf + 1) * 8)); HIP_TRY(B.get(&d_othead, (nf + 1) * 8)); HIP_TRY(B.get(&d_ottail, (nf + 1) * 8)); HIP_TRY(B.get(&d_oboff, (nf + 1) * 8));
  hipLaunchKernelGGL(ug_head_compact_kernel, dim3(ug_grid(nn)), dim3(UG_BLK), 0, s, d_hflag, d_hpos, d_fin_stamp, d_fin_thead, d_fin_ttail, d_clen, d_tail_out,
                     d_node_part, d_boff, nn, d_ostamp, d_othead, d_ottail, d_oclen, d_otail, d_opart, d_oboff);
  if (ne) hipLaunchKernelGGL(ug_edge_heads_kernel, dim3(ug_grid(ne)), dim3(UG_BLK), 0, s, d_esrc, d_edst, d_hpos, ne);
  std::vector<uint32_t> h_clen(nf), tail_out(nf), h_part(nf), h_esrc(ne), h_edst(ne);
  std::vector<uint64_t> h_stamp(nf), h_thead(nf), h_ttail(nf), h_boff(nf);
  std::string obases(nb, '\0');
  if (nf) {
    HIP_TRY(hipMemcpyAsync(h_clen.data(), d_oclen, nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(tail_out.data(), d_otail, nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_part.data(), d_opart, nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_stamp.data(), d_ostamp, nf * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_thead.data(), d_othead, nf * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_ttail.data(), d_ottail, nf * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h_boff.data(), d_oboff, nf * 8, hipMemcpyDeviceToHost, s));
  }
  if (nb) HIP_TRY(hipMemcpyAsync(&obases[0], d_obases, nb, hipMemcpyDeviceToHost, s));
  if (ne) { HIP_TRY(hipMemcpyAsync(h_esrc.data(), d_esrc, ne * 4, hipMemcpyDeviceToHost, s)); HIP_TRY(hipMemcpyAsync(h_edst.data(), d_edst, ne * 4, hipMemcpyDeviceToHost, s)); }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  lap("products + download");
  // ---- host: final nodes of every partition in creation order, edges with their list ranks (everything below is indexed
  // by compact head index: heads come partition by partition)
  std::vector<uint64_t> heads_of_part_off(n_parts + 1, 0);
  for (uint64_t i = 0; i < nf; i++) heads_of_part_off[h_part[i] + 1]++;
  for (uint32_t p = 0; p < n_parts; p++) heads_of_part_off[p + 1] += heads_of_part_off[p];
  std::vector<uint32_t> heads(nf);
  for (uint64_t i = 0; i < nf; i++) heads[i] = (uint32_t)i;
  std::vector<uint32_t> local_of(nf, UG_NONE);           // compact head index -> index among its partition's final nodes
  U->n_len.resize(nf); U->n_tail_out.resize(nf); U->base_off.assign(nf + 1, 0);
  U->bases.resize(nb);
  // (the partitions are independent: host threads take them in turn; where a partition's bases start follows from the lengths)
  std::vector<uint64_t> part_bases(n_parts + 1, 0);
  for (uint64_t i = 0; i < nf; i++) part_bases[h_part[i] + 1] += (uint64_t)K + h_clen[i] - 1;
  for (uint32_t p = 0; p < n_parts; p++) part_bases[p + 1] += part_bases[p];
  const unsigned n_host = (unsigned)std::max(1, std::min<int>(shn_host_cpus(), 16));
  auto over_parts = [&](const std::function<void(uint32_t)>& body) {
    std::atomic<uint32_t> next{0};
    auto work = [&]() { for (uint32_t p; (p = next.fetch_add(1)) < n_parts;) body(p); };
    if (n_host <= 1 || n_parts < 4 || nf < 4096) { work(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < std::min<unsigned>(n_host, n_parts); t++) th.emplace_back(work);
    for (auto& x : th) x.join();
  };
  over_parts([&](uint32_t p) {
    auto b = heads.begin() + heads_of_part_off[p], e = heads.begin() + heads_of_part_off[p + 1];
    std::sort(b, e, [&](uint32_t x, uint32_t y) { return h_stamp[x] < h_stamp[y]; });
    U->node_off[p] = heads_of_part_off[p];
    uint64_t bat = part_bases[p];
    for (auto it = b; it != e; ++it) {
      const uint64_t i = (uint64_t)(it - heads.begin());
      const uint32_t h = *it;
      local_of[h] = (uint32_t)(it - b);
      const uint64_t len = (uint64_t)K + h_clen[h] - 1;
      memcpy(&U->bases[bat], obases.data() + h_boff[h], len);
      U->base_off[i] = bat; bat += len;
      U->n_len[i] = h_clen[h]; U->n_tail_out[i] = tail_out[h];
    }
  });
  const uint64_t bat = part_bases[n_parts];
  U->node_off[n_parts] = nf;
  U->base_off[nf] = bat;
  // edges, grouped by partition (rows come partition by partition already), in edge-id order
  std::vector<uint64_t> eoff(n_parts + 1, 0);
  for (uint64_t x = 0; x < ne; x++) eoff[h_part[h_esrc[x]] + 1]++;
  for (uint32_t p = 0; p < n_parts; p++) eoff[p + 1] += eoff[p];
  U->edge_off = eoff;
  U->e_src.resize(ne); U->e_dst.resize(ne); U->e_out_rank.resize(ne); U->e_in_rank.resize(ne);
  std::vector<uint64_t> key_out(ne), key_in(ne);
  for (uint64_t x = 0; x < ne; x++) { key_out[x] = h_thead[h_edst[x]]; key_in[x] = h_ttail[h_esrc[x]]; }
  over_parts([&](uint32_t p) {
    const uint64_t lo = eoff[p], n = eoff[p + 1] - lo;
    if (!n) return;
    std::vector<uint32_t> idx;
    // list ranks (in row-index space): the edges of one source ordered by (key_out, row); of one destination by (key_in, row)
    std::vector<uint32_t> byo(n), byi(n), orank(n), irank(n);
    for (uint64_t i = 0; i < n; i++) byo[i] = byi[i] = (uint32_t)i;                      // row order
    std::stable_sort(byo.begin(), byo.end(), [&](uint32_t x, uint32_t y) {
      if (h_esrc[lo + x] != h_esrc[lo + y]) return h_esrc[lo + x] < h_esrc[lo + y];
      return key_out[lo + x] < key_out[lo + y]; });
    std::stable_sort(byi.begin(), byi.end(), [&](uint32_t x, uint32_t y) {
      if (h_edst[lo + x] != h_edst[lo + y]) return h_edst[lo + x] < h_edst[lo + y];
      return key_in[lo + x] < key_in[lo + y]; });
    for (uint64_t i = 0, r = 0; i < n; i++) { r = (i && h_esrc[lo + byo[i]] == h_esrc[lo + byo[i - 1]]) ? r + 1 : 0; orank[byo[i]] = (uint32_t)r; }
    for (uint64_t i = 0, r = 0; i < n; i++) { r = (i && h_edst[lo + byi[i]] == h_edst[lo + byi[i - 1]]) ? r + 1 : 0; irank[byi[i]] = (uint32_t)r; }
    // edge ids: by the merge that re-created the edge last.  A merge first re-creates the in-edges of its first node (in
    // in-list order), then the out-edges of its second node (in out-list order); edges never re-created keep row order, first.
    std::vector<uint64_t> ekey(n);
    std::vector<uint32_t> etie(n);
    for (uint64_t i = 0; i < n; i++) {
      const uint64_t a = key_out[lo + i] * 2, b = key_in[lo + i] ? key_in[lo + i] * 2 + 1 : 0;     // phase bit: head merge (in-edges) before tail absorption
      ekey[i] = std::max(a, b);
      etie[i] = ekey[i] == 0 ? (uint32_t)i : (a > b ? irank[i] : orank[i]);
    }
    idx.resize(n);
    for (uint64_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return ekey[x] != ekey[y] ? ekey[x] < ekey[y] : etie[x] < etie[y]; });
    for (uint64_t i = 0; i < n; i++) {
      const uint32_t x = idx[i];
      U->e_src[lo + i] = local_of[h_esrc[lo + x]]; U->e_dst[lo + i] = local_of[h_edst[lo + x]];
      U->e_out_rank[lo + i] = orank[x]; U->e_in_rank[lo + i] = irank[x];
    }
  });
  lap("host assembly");
  *out = U;
  guard.u = nullptr;
  return SHN_OK;
}
