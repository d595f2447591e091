// known_paths, the part every read goes through (mbgraph.py:1355-1388, known_paths -> search_sequence at :114-160): a read whose
// first K-mer occurs in node n at offset o and whose text equals the node's from there lies inside that node if it ends before
// the node does -- its only path is [n], it adds no known edge and no known path; otherwise it runs on into the node's
// successors and has to be searched.  Nearly every read is of the first kind (96 % at BASELINE configs[2]).  The host stage
// used to index every K-mer of every node (a 600 k-item sort per partition: 7 of the graph stage's 31 thread-seconds per
// step), look up every read's first and last K-mer on the device and compare every read with its node on host threads
// (4 thread-seconds).  Here the whole test runs where the reads already are:
//   kp_insert     one thread per base position of the concatenated node texts: its K-mer into an open-addressing table, a count
//                 per K-mer;  scan;  kp_fill: the positions of every K-mer side by side
//   kp_classify   per read: first / last K-mer looked up; the occurrences of the first K-mer by ascending position (= the order of
//                 the host's index: node order, then offset), text compared base by base against the node
//                 -> 0 (nothing to do), 1 + node (inside that node), 2 (search on the host), 3 + node + offset (search on the host
//                 from the K-mer's only occurrence)
// (a first version sorted the (K-mer, position) items: 35 launches per partition against 5)
// The host keeps the sequential part: reads of kind 2 and 3, in read order; those of kind 2 (rare: a K-mer occurs once in a de Bruijn
// graph until bridging copies nodes) against an index of just their first K-mers.
#include "common.h"
#include "graph_dev.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

namespace {

__device__ __forceinline__ int kp_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
__device__ __forceinline__ uint32_t kp_node_of(const uint64_t* __restrict__ off, uint32_t n_nodes, uint64_t p) {   // largest i with off[i] <= p
  uint32_t lo = 0, hi = n_nodes;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

// ---- node text packed like the reads (2 bits per base, MSB first in 64-bit words): a comparison of a read's tail with a node's text
// is then XORs of 32 bases at a time.  (Until round 6 kp_agree and the classification compared byte by byte -- a hundred dependent
// global byte loads per comparison, two depth-first runs per read: kp_search_all lasted 2.7 - 3.9 ms per partition at bench.py
// --config 2p, a kernel of a few thousand busy lanes that ends with its slowest thread.)
__global__ void kp_pack_kernel(const uint8_t* __restrict__ bases, uint64_t total, uint64_t* __restrict__ pk, uint64_t n_words) {
  const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= n_words) return;
  uint64_t x = 0;
  for (int q = 0; q < 32; q++) {
    const uint64_t p = wi * 32 + (uint64_t)q;
    const int c = p < total ? kp_code(bases[p]) : 0;
    x = (x << 2) | (uint64_t)(c < 0 ? 0 : c);                      // (a base outside ACGT is reported by kp_insert: the call fails)
  }
  pk[wi] = x;
}
// the 32 bases from position p on, MSB first (pk is padded with two words)
__device__ __forceinline__ uint64_t kp_get32(const uint64_t* __restrict__ pk, uint64_t p) {
  const uint64_t wi = p >> 5;
  const uint32_t sh = (uint32_t)(p & 31) * 2;
  const uint64_t a = pk[wi];
  return sh ? (a << sh) | (pk[wi + 1] >> (64 - sh)) : a;
}
// read[so .. so + n) == text[p .. p + n)?  (w: the read's words, padded by the read set's layout to whole words)
__device__ __forceinline__ bool kp_same(const uint64_t* __restrict__ w, uint32_t so, const uint64_t* __restrict__ pk, uint64_t p, uint32_t n) {
  for (uint32_t q = 0; q < n; q += 32) {
    const uint32_t m = min(32u, n - q);
    const uint32_t ro = so + q;
    const uint32_t wi = ro >> 5, sh = (ro & 31) * 2;
    uint64_t a = w[wi] << sh;
    if (sh && (ro & 31) + m > 32) a |= w[wi + 1] >> (64 - sh);      // (the second word only when the window reaches into it: no read past the read's words)
    const uint64_t x = (a ^ kp_get32(pk, p + q)) >> (64 - 2 * m);
    if (x) return false;
  }
  return true;
}

constexpr uint64_t KP_EMPTY = ~0ULL;                             // (a K-mer of K <= 31 bases is < 2^62)
__device__ __forceinline__ uint64_t kp_mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// one thread per base position of the concatenated node texts: the K-mer starting there (if it lies inside its node) enters an
// open-addressing table; slot_of[p] = its slot (the K-mer's group), cnt[slot] = occurrences
__global__ void kp_insert(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, uint32_t n_nodes, uint64_t total, int K,
                          unsigned long long* hkeys, uint64_t mask, uint32_t* __restrict__ cnt, uint32_t* __restrict__ slot_of,
                          unsigned long long* __restrict__ counters) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const uint32_t nd = kp_node_of(off, n_nodes, p);
  uint32_t slot = 0xFFFFFFFFu;
  if (p + (uint64_t)K <= off[nd + 1]) {
    uint64_t k = 0;
    bool ok = true;
    for (int j = 0; j < K; j++) { const int c = kp_code(bases[p + j]); if (c < 0) { ok = false; break; } k = (k << 2) | (uint64_t)c; }
    if (!ok) atomicAdd(&counters[1], 1ULL);
    else {
      uint64_t s = kp_mix(k) & mask;
      while (true) {
        unsigned long long cur = hkeys[s];
        if (cur == KP_EMPTY) { const unsigned long long old = atomicCAS(&hkeys[s], KP_EMPTY, (unsigned long long)k); cur = old == KP_EMPTY ? k : old; }
        if (cur == k) break;
        s = (s + 1) & mask;
      }
      slot = (uint32_t)s;
      atomicAdd(&cnt[s], 1u);
    }
  }
  slot_of[p] = slot;
}
// the occurrences of every K-mer side by side (in any order: the reader takes them by ascending position)
__global__ void kp_fill(const uint32_t* __restrict__ slot_of, uint64_t total, const uint64_t* __restrict__ goff, uint32_t* __restrict__ fill,
                        uint32_t* __restrict__ occ) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const uint32_t s = slot_of[p];
  if (s == 0xFFFFFFFFu) return;
  occ[goff[s] + atomicAdd(&fill[s], 1u)] = (uint32_t)p;
}

struct RView { const uint64_t* words; const uint64_t* woff; const uint32_t* len; uint64_t n; uint32_t fixed_len, wpr; };

__device__ __forceinline__ int64_t kp_find(const unsigned long long* __restrict__ hkeys, uint64_t mask, uint64_t key) {
  uint64_t s = kp_mix(key) & mask;
  while (true) {
    const unsigned long long cur = hkeys[s];
    if (cur == key) return (int64_t)s;
    if (cur == KP_EMPTY) return -1;
    s = (s + 1) & mask;
  }
}

__global__ void kp_classify(RView v, int K, const unsigned long long* __restrict__ hkeys, uint64_t mask, const uint64_t* __restrict__ goff,
                            const uint32_t* __restrict__ occ, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, uint32_t n_nodes,
                            uint8_t* __restrict__ state, int32_t* __restrict__ node_out, uint32_t* __restrict__ off_out,
                            int32_t* __restrict__ first_out, int32_t* __restrict__ last_out, uint32_t* __restrict__ slow_list = nullptr,
                            unsigned long long* __restrict__ slow_count = nullptr) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= v.n) return;
  const uint32_t L = v.len ? v.len[r] : v.fixed_len;
  uint8_t st = 0;
  int32_t fn = -1;
  uint32_t fo = 0;
  if (L >= (uint32_t)K) {
    const uint64_t* w = v.words + (v.woff ? v.woff[r] : r * v.wpr);
    const uint64_t kf = shn_extract(w, 0, K), kl = shn_extract(w, L - K, K);
    const int64_t sf = kp_find(hkeys, mask, kf);
    if (sf >= 0 && kp_find(hkeys, mask, kl) >= 0) {
      bool any = false, need = false;
      const uint64_t g0 = goff[sf], g1 = goff[sf + 1];
      int64_t last = -1;                                             // occurrences in index order = by ascending position
      for (uint64_t it = g0; it < g1 && !need; it++) {
        uint64_t p = ~0ULL;
        for (uint64_t j = g0; j < g1; j++) { const uint64_t q = occ[j]; if ((int64_t)q > last && q < p) p = q; }
        last = (int64_t)p;
        const uint32_t nd = kp_node_of(off, n_nodes, p);
        const uint64_t left = off[nd + 1] - p;                       // bases of the node from the occurrence on
        const uint32_t n = (uint32_t)min((uint64_t)L, left);
        bool same = true;
        for (uint32_t i = K; i < n && same; i++) {                   // (the first K bases are the K-mer itself)
          const uint32_t rb = (uint32_t)((w[i >> 5] >> (62 - 2 * (i & 31))) & 3ULL);
          same = kp_code(bases[p + i]) == (int)rb;
        }
        if (!same) continue;
        if ((uint64_t)L <= left) { fn = (int32_t)nd; any = true; }
        else {
          need = true;
          // the K-mer's only occurrence: the host starts its search right there, without an index of its own
          if (off_out && g1 - g0 == 1) { fn = (int32_t)nd; fo = (uint32_t)(p - off[nd]); }
        }
      }
      st = need ? (off_out && g1 - g0 == 1 ? 3 : 2) : any ? 1 : 0;
    }
  }
  state[r] = st;
  node_out[r] = fn;
  if (off_out) off_out[r] = fo;
  if (first_out) { first_out[r] = st == 1 ? fn : -1; last_out[r] = st == 1 ? fn : -1; }     // (a read inside one node: that node is its path)
  // the reads left to search (4 % of them), side by side for kp_search_all: a wavefront of 64 searches instead of 64 reads of which 2 search
  if (slow_list && st >= 2) slow_list[atomicAdd(slow_count, 1ULL)] = (uint32_t)r;
}


// ---- the search itself for the reads of state 3 (search_sequence, mbgraph.py:114-160 / mbgraph_host.hip): from the node and offset
// the read's first K-mer fixed, the read is followed node by node over the out-edges (in list order; an edge (dst, ov) is taken
// when the node text of dst from ov on agrees with the rest of the read), every complete way is a path.  One thread per read,
// depth first with an explicit stack -- twice: first counting the words its paths need, then, after ONE atomic reserved them,
// writing them: [read, length, nodes ...] records in the order the host's recursion would have produced them.  A read whose
// search runs deeper than KP_DEPTH nodes or finds no room keeps state 3 and is searched on the host as before.
#define KP_DEPTH 12
#define KP_HOPS 30
#define KP_DEEP (KP_HOPS + 1)  // levels of the LDS stack of kp_search_all: no search is cut short (the short nodes bridging leaves around repeats put a 100-base read on more than 12 of them)
struct KpGraph { const uint8_t* bases; const uint64_t* off; const uint32_t* eoff; const uint32_t* edst; const uint32_t* eov; const uint64_t* pk; };

__device__ __forceinline__ bool kp_agree(const uint64_t* __restrict__ w, uint32_t L, uint32_t so, const KpGraph& G, uint32_t node, uint32_t i) {
  // compare(seq, so, bases[node], i): the common length of the two tails
  const uint64_t nb = G.off[node + 1] - G.off[node];
  if ((uint64_t)i > nb) return false;
  const uint32_t n = (uint32_t)min((uint64_t)(L - so), nb - i);
  if (G.pk) return kp_same(w, so, G.pk, G.off[node] + i, n);
  const uint8_t* b = G.bases + G.off[node] + i;
  for (uint32_t q = 0; q < n; q++) {
    const uint32_t p = so + q;
    const uint32_t rb = (uint32_t)((w[p >> 5] >> (62 - 2 * (p & 31))) & 3ULL);
    if (kp_code(b[q]) != (int)rb) return false;
  }
  return true;
}

// one depth-first run; out == NULL: only counts.  Returns the words of all records, or -1 when the stack is too shallow.
// The stack: four words per level -- node, offset into the read, next edge, offset into the node -- either private arrays of KP_DEPTH
// levels (stk == NULL: kp_search) or `max_depth` levels in LDS, [word][level][lane] (kp_search_all: deep enough for every search, KP_HOPS
// + 1 levels; as private arrays that depth went through scratch memory -- 11 s of device time per step of bench.py --config 2p against 2).
template <bool LDS_STACK>
__device__ int kp_dfs_t(const uint64_t* __restrict__ w, uint32_t L, const KpGraph& G, uint32_t r, uint32_t node0, uint32_t i0, int32_t* __restrict__ out,
                        int32_t* last_end, uint32_t* stk, int max_depth, uint32_t stride) {
  uint32_t p_node[LDS_STACK ? 1 : KP_DEPTH], p_so[LDS_STACK ? 1 : KP_DEPTH], p_e[LDS_STACK ? 1 : KP_DEPTH], p_i[LDS_STACK ? 1 : KP_DEPTH];
#define ST_NODE(d) (LDS_STACK ? stk[(uint32_t)(0 * max_depth + (d)) * stride] : p_node[d])
#define ST_SO(d) (LDS_STACK ? stk[(uint32_t)(1 * max_depth + (d)) * stride] : p_so[d])
#define ST_E(d) (LDS_STACK ? stk[(uint32_t)(2 * max_depth + (d)) * stride] : p_e[d])
#define ST_I(d) (LDS_STACK ? stk[(uint32_t)(3 * max_depth + (d)) * stride] : p_i[d])
  if (!LDS_STACK) max_depth = KP_DEPTH;
  int depth = 0, words = 0;
  ST_NODE(0) = node0; ST_SO(0) = 0; ST_E(0) = 0xFFFFFFFFu;           // e = 0xFFFFFFFF: the node has just been entered
  ST_I(0) = i0;                                                       // offset into the node at which the read continues
  while (depth >= 0) {
    const uint32_t node = ST_NODE(depth);
    const uint32_t nl = (uint32_t)(G.off[node + 1] - G.off[node]) - ST_I(depth);
    if (ST_E(depth) == 0xFFFFFFFFu) {
      const int hops = KP_HOPS - depth;
      if (hops <= 0 || L - ST_SO(depth) <= nl) {                       // the read ends in this node (or the hop limit): a path
        if (out) { out[words] = (int32_t)r; out[words + 1] = depth + 1; for (int d = 0; d <= depth; d++) out[words + 2 + d] = (int32_t)ST_NODE(d); }
        if (last_end) *last_end = (int32_t)node;
        words += depth + 3;
        depth--;
        continue;
      }
      ST_E(depth) = G.eoff[node];
    }
    const uint32_t so2 = ST_SO(depth) + nl;
    bool went = false;
    while (ST_E(depth) < G.eoff[node + 1]) {
      const uint32_t e = ST_E(depth);
      ST_E(depth) = e + 1;
      const uint32_t dst = G.edst[e], ov = G.eov[e];
      if (!kp_agree(w, L, so2, G, dst, ov)) continue;
      if (depth + 1 >= max_depth) return -1;
      depth++;
      ST_NODE(depth) = dst; ST_SO(depth) = so2; ST_I(depth) = ov; ST_E(depth) = 0xFFFFFFFFu;
      went = true;
      break;
    }
    if (!went) depth--;
  }
  return words;
#undef ST_NODE
#undef ST_SO
#undef ST_E
#undef ST_I
}
__device__ __forceinline__ int kp_dfs(const uint64_t* __restrict__ w, uint32_t L, const KpGraph& G, uint32_t r, uint32_t node0, uint32_t i0, int32_t* __restrict__ out,
                                      int32_t* last_end = nullptr, uint32_t* stk = nullptr, int max_depth = 0, uint32_t stride = 0) {
  return stk ? kp_dfs_t<true>(w, L, G, r, node0, i0, out, last_end, stk, max_depth, stride)
             : kp_dfs_t<false>(w, L, G, r, node0, i0, out, last_end, nullptr, 0, 0);
}

__global__ void kp_search(RView v, KpGraph G, uint8_t* __restrict__ state, const int32_t* __restrict__ node_out, const uint32_t* __restrict__ off_out,
                          int32_t* __restrict__ paths, uint64_t cap, unsigned long long* __restrict__ cursor) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= v.n || state[r] != 3) return;
  const uint32_t L = v.len ? v.len[r] : v.fixed_len;
  const uint64_t* w = v.words + (v.woff ? v.woff[r] : r * v.wpr);
  const int need = kp_dfs(w, L, G, (uint32_t)r, (uint32_t)node_out[r], off_out[r], nullptr);
  if (need < 0) return;
  if (need == 0) { state[r] = 4; return; }                            // (no way through the graph: nothing to record)
  const unsigned long long at = atomicAdd(cursor, (unsigned long long)need);
  if (at + (unsigned long long)need > cap) return;                    // (no room: the host searches this read)
  kp_dfs(w, L, G, (uint32_t)r, (uint32_t)node_out[r], off_out[r], paths + at);
  state[r] = 4;
}


// ---- the device-resident form (graph_dev.h): state, node, offset, first / last node never leave the device; the reads whose first
// K-mer occurs more than once (state 2) are searched here too -- from every occurrence, in index order, as the host's loop does
struct KpIndex { const unsigned long long* hkeys; uint64_t mask; const uint64_t* goff; const uint32_t* occ; uint32_t n_nodes; int K; };

// all paths of read r (state 3: from its K-mer's only occurrence; state 2: from every occurrence by ascending position) -> words of
// their records (-1: a search ran too deep); first / last = the ends of the read's LAST path in that order, or the node the read lies
// inside at its last such occurrence (mbgraph.py:1379-1384 sets Read.nodes again with every path)
__device__ int kp_read_paths(const uint64_t* __restrict__ w, uint32_t L, const KpGraph& G, const KpIndex& X, uint32_t r, uint32_t st, uint32_t node0,
                             uint32_t off0, const uint8_t* __restrict__ bases, int32_t* __restrict__ out, int32_t& first, int32_t& last,
                             uint32_t* stk = nullptr, int max_depth = 0, uint32_t stride = 0) {
  first = -1; last = -1;
  if (st == 3) {
    int32_t le = -1;
    const int words = kp_dfs(w, L, G, r, node0, off0, out, &le, stk, max_depth, stride);
    if (words > 0) { first = (int32_t)node0; last = le; }
    return words;
  }
  const int64_t sf = kp_find(X.hkeys, X.mask, shn_extract(w, 0, X.K));
  if (sf < 0) return 0;
  const uint64_t g0 = X.goff[sf], g1 = X.goff[sf + 1];
  int words = 0;
  int64_t lastp = -1;
  for (uint64_t it = g0; it < g1; it++) {
    uint64_t p = ~0ULL;
    for (uint64_t j = g0; j < g1; j++) { const uint64_t q = X.occ[j]; if ((int64_t)q > lastp && q < p) p = q; }
    lastp = (int64_t)p;
    const uint32_t nd = kp_node_of(G.off, X.n_nodes, p);
    const uint64_t left = G.off[nd + 1] - p;
    const uint32_t n = (uint32_t)min((uint64_t)L, left);
    bool same = true;
    if (G.pk) same = n <= (uint32_t)X.K || kp_same(w, (uint32_t)X.K, G.pk, p + (uint64_t)X.K, n - (uint32_t)X.K);
    else
    for (uint32_t i = X.K; i < n && same; i++) {
      const uint32_t rb = (uint32_t)((w[i >> 5] >> (62 - 2 * (i & 31))) & 3ULL);
      same = kp_code(bases[p + i]) == (int)rb;
    }
    if (!same) continue;
    if ((uint64_t)L <= left) { first = (int32_t)nd; last = (int32_t)nd; continue; }
    int32_t le = -1;
    const int wds = kp_dfs(w, L, G, r, nd, (uint32_t)(p - G.off[nd]), out ? out + words : nullptr, &le, stk, max_depth, stride);
    if (wds < 0) return -1;
    if (wds > 0) { first = (int32_t)nd; last = le; }
    words += wds;
  }
  return words;
}

// counters: [0] words of records asked for, [1] nodes with a base outside ACGT (kp_insert), [2] reads with records, [3] reads left to the host
__global__ void kp_search_all(RView v, KpGraph G, KpIndex X, uint8_t* __restrict__ state, const int32_t* __restrict__ node_out,
                              const uint32_t* __restrict__ off_out, int32_t* __restrict__ paths, uint64_t cap, unsigned long long* __restrict__ counters,
                              int32_t* __restrict__ d_first, int32_t* __restrict__ d_last, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ rec_cnt,
                              uint64_t rec_cap, KpSlow* __restrict__ left, uint64_t left_cap, const uint32_t* __restrict__ slow_list,
                              const unsigned long long* __restrict__ slow_count) {
  const uint64_t n_slow = *slow_count;
  __shared__ uint32_t kp_stack[4 * KP_DEEP * 64];                    // the depth-first stacks of the block's 64 lanes (blockDim.x == 64)
  uint32_t* const stk = kp_stack + threadIdx.x;
  for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n_slow; it += (uint64_t)gridDim.x * blockDim.x) {
  const uint64_t r = slow_list[it];
  const uint32_t st = state[r];
  if (st != 2 && st != 3) continue;
  const uint32_t L = v.len ? v.len[r] : v.fixed_len;
  const uint64_t* w = v.words + (v.woff ? v.woff[r] : r * v.wpr);
  const uint32_t node0 = (uint32_t)node_out[r], off0 = off_out[r];
  int32_t f, l;
  const int need = kp_read_paths(w, L, G, X, (uint32_t)r, st, node0, off0, G.bases, nullptr, f, l, stk, KP_DEEP, 64);
  bool done = need == 0;
  if (need > 0) {
    const unsigned long long at = atomicAdd(&counters[0], (unsigned long long)need);
    if (at + (unsigned long long)need <= cap) {
      const unsigned long long q = atomicAdd(&counters[2], 1ULL);
      if (q < rec_cap) {
        kp_read_paths(w, L, G, X, (uint32_t)r, st, node0, off0, G.bases, paths + at, f, l, stk, KP_DEEP, 64);
        rec_cnt[2 * q] = (uint32_t)r; rec_cnt[2 * q + 1] = cnt[r];
        done = true;
      }
    }
  }
  if (done) { state[r] = 4; d_first[r] = f; d_last[r] = l; continue; }
  const unsigned long long q = atomicAdd(&counters[3], 1ULL);
  if (q < left_cap) left[q] = KpSlow{(uint32_t)r, st, node0, off0, cnt[r]};
  }
}

__global__ void kp_patch(const int32_t* __restrict__ patches, uint64_t n, int32_t* __restrict__ first, int32_t* __restrict__ last) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = patches[3 * i];
  first[r] = patches[3 * i + 1]; last[r] = patches[3 * i + 2];
}
// find_mate_pairs' pass over the reads (mbgraph.py:839-880 / mbgraph_host.hip): first mates whose last node and whose mate's first node
// differ and are not joined by an edge; every distinct (a, b) once (a hash set; counters[1] set: the set overflowed)
__global__ void kp_mates(uint64_t nr, const int32_t* __restrict__ mate, const uint8_t* __restrict__ role, const int32_t* __restrict__ first,
                         const int32_t* __restrict__ last, const uint32_t* __restrict__ eoff, const uint32_t* __restrict__ edst,
                         unsigned long long* __restrict__ set, uint64_t mask, uint32_t* __restrict__ pairs, uint64_t cap, unsigned long long* __restrict__ counters) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nr || role[r] != 1) return;
  const int32_t m = mate[r];
  if (m < 0) return;
  const int32_t a = last[r], b = first[m];
  if (a < 0 || first[r] < 0 || b < 0 || last[m] < 0 || a == b) return;
  for (uint32_t e = eoff[a]; e < eoff[a + 1]; e++) if (edst[e] == (uint32_t)b) return;
  const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | (uint32_t)b;
  uint64_t s = kp_mix(key) & mask;
  for (int probe = 0; probe < 4096; probe++) {
    unsigned long long cur = set[s];
    if (cur == KP_EMPTY) { const unsigned long long old = atomicCAS(&set[s], KP_EMPTY, key); cur = old == KP_EMPTY ? KP_EMPTY : old; if (old == KP_EMPTY) {
        const unsigned long long q = atomicAdd(&counters[0], 1ULL);
        if (q < cap) { pairs[2 * q] = (uint32_t)a; pairs[2 * q + 1] = (uint32_t)b; } else atomicAdd(&counters[1], 1ULL);
        return; } }
    if (cur == key) return;
    s = (s + 1) & mask;
  }
  atomicAdd(&counters[1], 1ULL);
}

}  // namespace

// reads: the distinct reads of the partition (ACGT only); node_bases / node_off: the texts of the partition's nodes one after
// the other, in the order the host's seed index would list them; state_out[r] / node_out[r] as described above (node = index
// into that order).  SHN_ERR_ARG if a node holds a base outside ACGT or K > 31.
static int kp_scan_impl(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off,
                        uint64_t n_nodes, uint8_t* state_out, int32_t* node_out, uint32_t* offset_out,
                        const uint32_t* edge_off, const uint32_t* edge_dst, const uint32_t* edge_ov, int32_t* paths_out, uint64_t paths_cap, uint64_t* paths_used) {
  if (paths_used) *paths_used = 0;
  if (!ctx || !reads || !node_off || (n_nodes && !node_bases) || (reads->n_reads && (!state_out || !node_out)))
    return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: NULL argument");
  if (K < 1 || K > 31) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: K must be in [1,31]");
  if (reads->n_invalid) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: reads contain non-ACGT bases");
  const uint64_t nr = reads->n_reads, total = n_nodes ? node_off[n_nodes] : 0;
  if (!nr) return SHN_OK;
  if (!total || n_nodes >= 0x7FFFFFFFULL || total >= 0xFFFFFFFFULL) {
    if (total) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: too many nodes / bases");
    std::fill(state_out, state_out + nr, (uint8_t)0);
    for (uint64_t i = 0; i < nr; i++) node_out[i] = -1;
    if (offset_out) std::fill(offset_out, offset_out + nr, 0u);
    return SHN_OK;
  }
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_SEEDS);
  ShnDevBufs bufs(s);
  uint64_t T = 1024;
  while (T < 2 * total) T <<= 1;
  uint8_t *d_bases = nullptr, *d_state = nullptr;
  uint64_t *d_off = nullptr, *d_goff = nullptr;
  unsigned long long *d_hkeys = nullptr, *d_cnt2 = nullptr;
  uint32_t *d_cnt = nullptr, *d_fill = nullptr, *d_slot = nullptr, *d_occ = nullptr;
  int32_t* d_node = nullptr;
  uint32_t* d_ofs = nullptr;
#define TRYK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_known_paths_scan: ") + hipGetErrorString(e_)); } while (0)
  TRYK(bufs.get(&d_bases, total + 8));
  TRYK(bufs.get(&d_off, (n_nodes + 1) * 8));
  TRYK(bufs.get(&d_hkeys, T * 8));
  TRYK(bufs.get(&d_cnt, (T + 1) * 4)); TRYK(bufs.get(&d_fill, (T + 1) * 4));
  TRYK(bufs.get(&d_goff, (T + 2) * 8));
  TRYK(bufs.get(&d_slot, (total + 1) * 4)); TRYK(bufs.get(&d_occ, (total + 1) * 4));
  TRYK(bufs.get(&d_state, nr + 1)); TRYK(bufs.get(&d_node, (nr + 1) * 4));
  TRYK(bufs.get(&d_cnt2, 16));
  if (offset_out) TRYK(bufs.get(&d_ofs, (nr + 1) * 4));
  TRYK(hipMemcpyAsync(d_bases, node_bases, total, hipMemcpyHostToDevice, s));
  TRYK(hipMemcpyAsync(d_off, node_off, (n_nodes + 1) * 8, hipMemcpyHostToDevice, s));
  TRYK(hipMemsetAsync(d_cnt2, 0, 16, s));
  TRYK(hipMemsetAsync(d_hkeys, 0xFF, T * 8, s));
  TRYK(hipMemsetAsync(d_cnt, 0, (T + 1) * 4, s));
  TRYK(hipMemsetAsync(d_fill, 0, (T + 1) * 4, s));
  const uint32_t gp = (uint32_t)cdiv(total, 256);
  hipLaunchKernelGGL(kp_insert, dim3(gp), dim3(256), 0, s, d_bases, d_off, (uint32_t)n_nodes, total, K, d_hkeys, T - 1, d_cnt, d_slot, d_cnt2);
  { int rc = shn_device_scan_u32(ctx, d_cnt, T, d_goff, nullptr); if (rc) return rc; }
  hipLaunchKernelGGL(kp_fill, dim3(gp), dim3(256), 0, s, d_slot, total, d_goff, d_fill, d_occ);
  RView v{reads->d_words, reads->d_woff, reads->d_len, nr, reads->fixed_len, reads->wpr};
  hipLaunchKernelGGL(kp_classify, dim3((uint32_t)cdiv(nr, 256)), dim3(256), 0, s, v, K, d_hkeys, T - 1, d_goff, d_occ, d_bases, d_off, (uint32_t)n_nodes,
                     d_state, d_node, offset_out ? d_ofs : nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  TRYK(hipGetLastError());
  int32_t* d_paths = nullptr;
  const bool search = offset_out && edge_off && paths_out && paths_cap && paths_used;
  if (search) {
    const uint64_t ne = edge_off[n_nodes];
    uint32_t *d_eoff = nullptr, *d_edst = nullptr, *d_eov = nullptr;
    TRYK(bufs.get(&d_eoff, (n_nodes + 1) * 4)); TRYK(bufs.get(&d_edst, (ne + 1) * 4)); TRYK(bufs.get(&d_eov, (ne + 1) * 4));
    TRYK(bufs.get(&d_paths, (paths_cap + 1) * 4));
    TRYK(hipMemsetAsync(d_paths, 0, (paths_cap + 1) * 4, s));             // (a reader walks the records by their lengths and stops at a length of 0)
    TRYK(hipMemcpyAsync(d_eoff, edge_off, (n_nodes + 1) * 4, hipMemcpyHostToDevice, s));
    if (ne) { TRYK(hipMemcpyAsync(d_edst, edge_dst, ne * 4, hipMemcpyHostToDevice, s)); TRYK(hipMemcpyAsync(d_eov, edge_ov, ne * 4, hipMemcpyHostToDevice, s)); }
    KpGraph G{d_bases, d_off, d_eoff, d_edst, d_eov, nullptr};
    hipLaunchKernelGGL(kp_search, dim3((uint32_t)cdiv(nr, 64)), dim3(64), 0, s, v, G, d_state, (const int32_t*)d_node, (const uint32_t*)d_ofs, d_paths, paths_cap,
                       d_cnt2 + 0);
    TRYK(hipGetLastError());
  }
  unsigned long long cnt[2] = {0, 0};
  TRYK(hipMemcpyAsync(cnt, d_cnt2, 16, hipMemcpyDeviceToHost, s));
  TRYK(hipMemcpyAsync(state_out, d_state, nr, hipMemcpyDeviceToHost, s));
  TRYK(hipMemcpyAsync(node_out, d_node, nr * 4, hipMemcpyDeviceToHost, s));
  if (offset_out) TRYK(hipMemcpyAsync(offset_out, d_ofs, nr * 4, hipMemcpyDeviceToHost, s));
  TRYK(hipStreamSynchronize(s));
  if (cnt[1]) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: a node holds a base outside ACGT");
  if (search) {
    // (the cursor counts what was asked for: once a request does not fit, none after it does; the records lie below it, the buffer
    // was zeroed, so the reader stops at the first record of length 0)
    const uint64_t used = std::min<uint64_t>(cnt[0], paths_cap);
    if (used) TRYK(hipMemcpy(paths_out, d_paths, used * 4, hipMemcpyDeviceToHost));
    *paths_used = used;
  }
#undef TRYK
  return SHN_OK;
}

extern "C" int shn_known_paths_scan(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off,
                                    uint64_t n_nodes, uint8_t* state_out, int32_t* node_out, uint32_t* offset_out) {
  return kp_scan_impl(ctx, reads, K, node_bases, node_off, n_nodes, state_out, node_out, offset_out, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
}

// The same, and the reads of state 3 searched on the device (kp_search): edge_off[n_nodes + 1] / edge_dst / edge_ov = the out-edges of
// the nodes in list order (destination as an index into the given node order, offset into the destination at which it continues
// the source: search_sequence's `ew`).  A read searched there gets state 4; its paths are records [read, length, node indices ...]
// in paths_out (paths_cap words), in the order search_sequence enumerates them, the records of one read in ascending position; a
// reader walks the records by their lengths up to `*paths_used` words and stops at a length of 0 (a request that found no room).
extern "C" int shn_known_paths_search(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off, uint64_t n_nodes,
                                      const uint32_t* edge_off, const uint32_t* edge_dst, const uint32_t* edge_ov, uint8_t* state_out, int32_t* node_out,
                                      uint32_t* offset_out, int32_t* paths_out, uint64_t paths_cap, uint64_t* paths_used) {
  if (!edge_off || !offset_out || !paths_out || !paths_used) return shn_fail(SHN_ERR_ARG, "shn_known_paths_search: NULL argument");
  return kp_scan_impl(ctx, reads, K, node_bases, node_off, n_nodes, state_out, node_out, offset_out, edge_off, edge_dst, edge_ov, paths_out, paths_cap, paths_used);
}

// ---- graph_dev.h: known_paths with its per-read state left on the device
void shn_kp_destroy(shn_kp* kp) {
  if (!kp) return;
  if (kp->ctx) hipSetDevice(kp->ctx->device);
  shn_dev_free(kp->d_first); shn_dev_free(kp->d_last); shn_dev_free(kp->d_eoff); shn_dev_free(kp->d_edst);
  delete kp;
}

int shn_known_paths_dev(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off, uint64_t n_nodes,
                        const uint32_t* edge_off, const uint32_t* edge_dst, const uint32_t* edge_ov, const shn_dedup* dd, std::vector<KpSlow>* left,
                        std::vector<int32_t>* records, std::vector<uint32_t>* rec_cnt, shn_kp** kp_out) {
  if (!ctx || !reads || !node_off || !edge_off || !dd || !left || !records || !rec_cnt || !kp_out || (n_nodes && !node_bases))
    return shn_fail(SHN_ERR_ARG, "shn_known_paths_dev: NULL argument");
  if (K < 1 || K > 31) return shn_fail(SHN_ERR_ARG, "shn_known_paths_dev: K must be in [1,31]");
  if (reads->n_invalid) return shn_fail(SHN_ERR_ARG, "shn_known_paths_dev: reads contain non-ACGT bases");
  const uint64_t nr = reads->n_reads, total = n_nodes ? node_off[n_nodes] : 0;
  if (nr != dd->n_distinct) return shn_fail(SHN_ERR_ARG, "shn_known_paths_dev: the reads are not the distinct reads of the duplicate search");
  left->clear(); records->clear(); rec_cnt->clear();
  *kp_out = nullptr;
  if (!nr || !total || n_nodes >= 0x7FFFFFFFULL || total >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_known_paths_dev: no reads / no nodes / too many bases");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_SEEDS);
  ShnDevBufs bufs(s);
  uint64_t T = 1024;
  while (T < 2 * total) T <<= 1;
  const uint64_t ne = edge_off[n_nodes];
  // room for the records [read, length, nodes ...] of the searched reads.  (Until round 6 it was nr / 2 words: at BASELINE configs[2] 4 % of
  // the reads run past their first node and a record is ~7 words; in the partitions of bench.py --config 2p -- short repeat nodes, 8 % of
  // the reads run on -- the records of 440 k reads need 230 k words, the room was 224 k, and every read behind the overflow went to the
  // host's recursive search: 16,000 reads, 31 ms per partition, 12 of the graph stage's 36 thread-seconds per step.)
  const uint64_t paths_cap = std::max<uint64_t>(1u << 18, 2 * nr + 4096), rec_cap = paths_cap / 3 + 1, left_cap = std::max<uint64_t>(1u << 16, nr / 8);
  shn_kp* kp = new shn_kp();
  kp->ctx = ctx; kp->n_reads = nr; kp->n_nodes = n_nodes;
  uint8_t *d_bases = nullptr, *d_state = nullptr;
  uint64_t *d_off = nullptr, *d_goff = nullptr;
  unsigned long long *d_hkeys = nullptr, *d_cnt2 = nullptr;
  uint32_t *d_cnt = nullptr, *d_fill = nullptr, *d_slot = nullptr, *d_occ = nullptr, *d_ofs = nullptr, *d_eov = nullptr, *d_rec_cnt = nullptr;
  int32_t *d_node = nullptr, *d_paths = nullptr;
  KpSlow* d_left = nullptr;
#define TRYK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { shn_kp_destroy(kp); return shn_fail(SHN_ERR_HIP, std::string("shn_known_paths_dev: ") + hipGetErrorString(e_)); } } while (0)
  TRYK(shn_dev_malloc(&kp->d_first, (nr + 1) * 4)); TRYK(shn_dev_malloc(&kp->d_last, (nr + 1) * 4));
  TRYK(shn_dev_malloc(&kp->d_eoff, (n_nodes + 1) * 4)); TRYK(shn_dev_malloc(&kp->d_edst, (ne + 1) * 4));
  TRYK(bufs.get(&d_bases, total + 8));
  TRYK(bufs.get(&d_off, (n_nodes + 1) * 8));
  TRYK(bufs.get(&d_hkeys, T * 8));
  TRYK(bufs.get(&d_cnt, (T + 1) * 4)); TRYK(bufs.get(&d_fill, (T + 1) * 4));
  TRYK(bufs.get(&d_goff, (T + 2) * 8));
  TRYK(bufs.get(&d_slot, (total + 1) * 4)); TRYK(bufs.get(&d_occ, (total + 1) * 4));
  TRYK(bufs.get(&d_state, nr + 1)); TRYK(bufs.get(&d_node, (nr + 1) * 4)); TRYK(bufs.get(&d_ofs, (nr + 1) * 4));
  TRYK(bufs.get(&d_cnt2, 64));
  uint32_t* d_slow = nullptr;
  TRYK(bufs.get(&d_slow, (nr + 1) * 4));
  TRYK(bufs.get(&d_eov, (ne + 1) * 4));
  TRYK(bufs.get(&d_paths, (paths_cap + 1) * 4)); TRYK(bufs.get(&d_rec_cnt, (rec_cap + 1) * 8)); TRYK(bufs.get(&d_left, (left_cap + 1) * sizeof(KpSlow)));
  TRYK(hipMemcpyAsync(d_bases, node_bases, total, hipMemcpyHostToDevice, s));
  TRYK(hipMemcpyAsync(d_off, node_off, (n_nodes + 1) * 8, hipMemcpyHostToDevice, s));
  TRYK(hipMemcpyAsync(kp->d_eoff, edge_off, (n_nodes + 1) * 4, hipMemcpyHostToDevice, s));
  if (ne) { TRYK(hipMemcpyAsync(kp->d_edst, edge_dst, ne * 4, hipMemcpyHostToDevice, s)); TRYK(hipMemcpyAsync(d_eov, edge_ov, ne * 4, hipMemcpyHostToDevice, s)); }
  TRYK(hipMemsetAsync(d_cnt2, 0, 64, s));
  TRYK(hipMemsetAsync(d_hkeys, 0xFF, T * 8, s));
  TRYK(hipMemsetAsync(d_cnt, 0, (T + 1) * 4, s));
  TRYK(hipMemsetAsync(d_fill, 0, (T + 1) * 4, s));
  TRYK(hipMemsetAsync(d_paths, 0, (paths_cap + 1) * 4, s));            // (a reader walks the records by their lengths and stops at a length of 0)
  const uint32_t gp = (uint32_t)cdiv(total, 256);
  hipLaunchKernelGGL(kp_insert, dim3(gp), dim3(256), 0, s, d_bases, d_off, (uint32_t)n_nodes, total, K, d_hkeys, T - 1, d_cnt, d_slot, d_cnt2);
  { int rc = shn_device_scan_u32(ctx, d_cnt, T, d_goff, nullptr); if (rc) { shn_kp_destroy(kp); return rc; } }
  hipLaunchKernelGGL(kp_fill, dim3(gp), dim3(256), 0, s, d_slot, total, d_goff, d_fill, d_occ);
  RView v{reads->d_words, reads->d_woff, reads->d_len, nr, reads->fixed_len, reads->wpr};
  { TimerRegion tcl(ctx, T_KP_CLASSIFY);
    // per distinct read: its packed words, two K-mer probes (slot 8 + occurrence list 8 each), the node text it is compared with (a byte per base), state / node / offset / first / last written (17)
    tcl.bytes(nr * ((uint64_t)reads->wpr * 8 + 32 + (reads->fixed_len ? reads->fixed_len : reads->max_len) + 17));
    hipLaunchKernelGGL(kp_classify, dim3((uint32_t)cdiv(nr, 256)), dim3(256), 0, s, v, K, d_hkeys, T - 1, d_goff, d_occ, d_bases, d_off, (uint32_t)n_nodes,
                       d_state, d_node, d_ofs, kp->d_first, kp->d_last, d_slow, d_cnt2 + 4); }
  uint64_t* d_pk = nullptr;
  const uint64_t pk_words = (total + 31) / 32;
  TRYK(bufs.get(&d_pk, (pk_words + 2) * 8));
  TRYK(hipMemsetAsync(d_pk + pk_words, 0, 16, s));
  hipLaunchKernelGGL(kp_pack_kernel, dim3((uint32_t)cdiv(pk_words, 256)), dim3(256), 0, s, (const uint8_t*)d_bases, total, d_pk, pk_words);
  KpGraph G{d_bases, d_off, kp->d_eoff, kp->d_edst, d_eov, (getenv("SHN_KP_PACKED") && getenv("SHN_KP_PACKED")[0] == '0') ? (const uint64_t*)nullptr : (const uint64_t*)d_pk};
  KpIndex X{d_hkeys, T - 1, d_goff, d_occ, (uint32_t)n_nodes, K};
  // (kp_insert's count of bad nodes sits in d_cnt2[1]; the search keeps its own counters in [0], [2], [3])
  { TimerRegion tse(ctx, T_KP_SEARCH);
    hipLaunchKernelGGL(kp_search_all, dim3((uint32_t)std::min<uint64_t>(cdiv(nr, 64), 16384)), dim3(64), 0, s, v, G, X, d_state, (const int32_t*)d_node, (const uint32_t*)d_ofs, d_paths, paths_cap,
                       d_cnt2, kp->d_first, kp->d_last, (const uint32_t*)dd->d_cnt, d_rec_cnt, rec_cap, d_left, left_cap, (const uint32_t*)d_slow,
                       (const unsigned long long*)(d_cnt2 + 4)); }
  TRYK(hipGetLastError());
  unsigned long long cnt[4] = {0, 0, 0, 0};
  TRYK(hipMemcpyAsync(cnt, d_cnt2, 32, hipMemcpyDeviceToHost, s));
  TRYK(hipStreamSynchronize(s));
  // (the search's bytes, known afterwards: per searched read its packed words twice -- the counting run and the writing run --, ~2 node
  // texts of its length per run at 2 bits a base, its record words written)
  if (ctx->timing) __atomic_fetch_add(&ctx->abytes[T_KP_SEARCH], (cnt[2] + cnt[3]) * (2 * (uint64_t)reads->wpr * 8 + (uint64_t)(reads->fixed_len ? reads->fixed_len : reads->max_len)) + cnt[0] * 4, __ATOMIC_RELAXED);
  if (cnt[1]) { shn_kp_destroy(kp); return shn_fail(SHN_ERR_ARG, "shn_known_paths_dev: a node holds a base outside ACGT"); }
  if (cnt[3] > left_cap) { shn_kp_destroy(kp); return shn_fail(SHN_ERR_INTERNAL, "shn_known_paths_dev: more reads left to the host than expected"); }
  const uint64_t used = std::min<uint64_t>(cnt[0], paths_cap), n_rec = std::min<uint64_t>(cnt[2], rec_cap);
  records->resize(used); rec_cnt->resize(2 * n_rec); left->resize(cnt[3]);
  if (used) TRYK(hipMemcpyAsync(records->data(), d_paths, used * 4, hipMemcpyDeviceToHost, s));
  if (n_rec) TRYK(hipMemcpyAsync(rec_cnt->data(), d_rec_cnt, n_rec * 8, hipMemcpyDeviceToHost, s));
  if (cnt[3]) TRYK(hipMemcpyAsync(left->data(), d_left, cnt[3] * sizeof(KpSlow), hipMemcpyDeviceToHost, s));
  TRYK(hipStreamSynchronize(s));
#undef TRYK
  *kp_out = kp;
  return SHN_OK;
}

int shn_kp_mate_pairs(shn_kp* kp, const shn_dedup* dd, const int32_t* patches, uint64_t n_patches, std::vector<uint32_t>* pairs_out) {
  if (!kp || !dd || !pairs_out || (n_patches && !patches)) return shn_fail(SHN_ERR_ARG, "shn_kp_mate_pairs: NULL argument");
  pairs_out->clear();
  if (!dd->paired || !kp->n_reads) return SHN_OK;
  shn_ctx* ctx = kp->ctx;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  ShnDevBufs bufs(s);
  uint64_t T = 1u << 16;
  while (T < 64 * kp->n_nodes && T < (1u << 24)) T <<= 1;
  const uint64_t cap = T / 2;
  int32_t* d_patch = nullptr; unsigned long long *d_set = nullptr, *d_cnt = nullptr; uint32_t* d_pairs = nullptr;
#define TRYM(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_kp_mate_pairs: ") + hipGetErrorString(e_)); } while (0)
  TRYM(bufs.get(&d_set, T * 8)); TRYM(bufs.get(&d_cnt, 16)); TRYM(bufs.get(&d_pairs, (cap + 1) * 8));
  TRYM(hipMemsetAsync(d_set, 0xFF, T * 8, s)); TRYM(hipMemsetAsync(d_cnt, 0, 16, s));
  if (n_patches) {
    for (uint64_t i = 0; i < n_patches; i++) if (patches[3 * i] < 0 || (uint64_t)patches[3 * i] >= kp->n_reads) return shn_fail(SHN_ERR_ARG, "shn_kp_mate_pairs: patch of a read that does not exist");
    TRYM(bufs.get(&d_patch, n_patches * 12));
    TRYM(hipMemcpyAsync(d_patch, patches, n_patches * 12, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kp_patch, dim3((uint32_t)cdiv(n_patches, 256)), dim3(256), 0, s, (const int32_t*)d_patch, n_patches, kp->d_first, kp->d_last);
  }
  hipLaunchKernelGGL(kp_mates, dim3((uint32_t)cdiv(kp->n_reads, 256)), dim3(256), 0, s, kp->n_reads, (const int32_t*)dd->d_mate, (const uint8_t*)dd->d_role,
                     (const int32_t*)kp->d_first, (const int32_t*)kp->d_last, (const uint32_t*)kp->d_eoff, (const uint32_t*)kp->d_edst, d_set, T - 1, d_pairs, cap, d_cnt);
  TRYM(hipGetLastError());
  unsigned long long cnt[2] = {0, 0};
  TRYM(hipMemcpyAsync(cnt, d_cnt, 16, hipMemcpyDeviceToHost, s));
  TRYM(hipStreamSynchronize(s));
  if (cnt[1]) return shn_fail(SHN_ERR_INTERNAL, "shn_kp_mate_pairs: more distinct node pairs than the set holds");
  pairs_out->resize(2 * cnt[0]);
  if (cnt[0]) { TRYM(hipMemcpyAsync(pairs_out->data(), d_pairs, cnt[0] * 8, hipMemcpyDeviceToHost, s)); TRYM(hipStreamSynchronize(s)); }
#undef TRYM
  return SHN_OK;
}
