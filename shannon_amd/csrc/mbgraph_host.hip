// Native host implementation of the multibridged de-Bruijn graph of one partition (rows a12-a24):
// the same algorithm as shannon_amd/mbgraph.py (itself the index-based mirror of the reference's
// multibridging.py:145-325 + mbgraph.py), in C++ because at realistic coverage the per-partition
// graph surgery in Python was 2/3 of the whole step.  Control logic over small irregular graphs:
// host code; the order-defining conventions (P1-P3) are identical to the Python mirror,
// which stays as the readable specification and is cross-checked against this in the tests.
#include "common.h"
#include <atomic>
#include "flatmap.h"
#include "graph_dev.h"
#include "unitigs.h"
#include "graph_result.h"
#include <thread>
#include <mutex>
#include <sys/mman.h>
#include <condition_variable>
#include <string>
#include <vector>
#include <unordered_map>
#include <map>
#include <set>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>

extern "C" int shn_known_paths_scan(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off,
                                    uint64_t n_nodes, uint8_t* state_out, int32_t* node_out, uint32_t* offset_out);

namespace {

typedef std::pair<int, int> RI;   // (read id, index)

static inline int base_code(char c) {
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// unique packed K-mer keys -> groups of occurrences (kept in insertion order inside a group)
struct SeedIndex {
  std::vector<uint64_t> keys;               // sorted unique
  std::vector<uint32_t> goff;               // CSR into occ
  std::vector<std::pair<int, int>> occ;     // (node, offset)
  void build(std::vector<std::pair<uint64_t, std::pair<int, int>>>& items) {
    std::stable_sort(items.begin(), items.end(), [](const std::pair<uint64_t, std::pair<int, int>>& a, const std::pair<uint64_t, std::pair<int, int>>& b) { return a.first < b.first; });
    keys.clear(); goff.clear(); occ.clear();
    for (size_t i = 0; i < items.size(); i++) {
      if (i == 0 || items[i].first != items[i - 1].first) { keys.push_back(items[i].first); goff.push_back((uint32_t)occ.size()); }
      occ.push_back(items[i].second);
    }
    goff.push_back((uint32_t)occ.size());
  }
  int find(uint64_t key) const {
    auto it = std::lower_bound(keys.begin(), keys.end(), key);
    return (it != keys.end() && *it == key) ? (int)(it - keys.begin()) : -1;
  }
};

// read string living in the interner arena
struct RStr {
  const char* p; size_t n;
  size_t size() const { return n; }
  const char& operator[](size_t i) const { return p[i]; }
  int compare(size_t pos, size_t len, const std::string& o) const {
    size_t m = std::min(len, n - pos);
    int c = memcmp(p + pos, o.data(), std::min(m, o.size()));
    if (c) return c;
    return m < o.size() ? -1 : (m > o.size() ? 1 : 0);
  }
  std::string substr(size_t pos, size_t len) const { return std::string(p + pos, std::min(len, n - pos)); }
  const char* begin() const { return p; }
  const char* end() const { return p + n; }
};

// ascending order for a list that is usually ascending already or two ascending runs one after the other (a bridging step hands a
// new node the reads of the old one twice, each time in order): a merge instead of a sort
template <class V> static void sort_runs(V& v) {
  auto b = v.begin(), e = v.end();
  auto p = std::is_sorted_until(b, e);
  if (p == e) return;
  if (std::is_sorted(p, e)) { std::inplace_merge(b, p, e); return; }
  std::sort(b, e);
}

// several partitions run on host threads at once; every thread has a context / stream of its own (ThreadCtx below), the
// calls its GPU sections make use per-call or per-context buffers only, so the sections overlap on the device

// Host-thread budget shared by the partitions of a process.  The multi-threaded phases of a partition (read decode, duplicate
// search, numbering, path classification) take as many tokens as they start threads; with 64 partitions beginning at once and up
// to 32 threads each the cores were oversubscribed five-fold and every phase ran ten times slower than alone -- the largest
// partition, which bounds the stage, included.  Partitions are started largest first, so the large ones get their threads first.
struct ThreadBudget {
  std::mutex mu; std::condition_variable cv; int avail;
  ThreadBudget() { const int hw = shn_host_cpus(); avail = std::max(4, hw); if (getenv("SHN_GRAPH_HOST_THREADS")) avail = std::max(1, atoi(getenv("SHN_GRAPH_HOST_THREADS"))); total = avail; }
  int total;
  // large requests (the large partitions, which bound the stage) never wait -- they may overdraw the budget; small ones wait for it
  void acquire(int n) { n = std::min(n, total); std::unique_lock<std::mutex> lk(mu); if (n < 8) cv.wait(lk, [&] { return avail >= n; }); avail -= n; }
  void release(int n) { n = std::min(n, total); { std::lock_guard<std::mutex> lk(mu); avail += n; } cv.notify_all(); }
  // as many of the n wanted as are free right now (at least 1: the caller's own thread), without waiting and without overdrawing:
  // for phases that are worth spreading only when the machine is otherwise idle (the last, largest partition of a stage)
  // (the partitions that are running right now each keep a core busy themselves: `others`)
  int take_free(int n, int others) { std::lock_guard<std::mutex> lk(mu); const int got = std::max(1, std::min(n, avail - others)); avail -= got; return got; }
};
static std::atomic<int> g_partitions_running{0};      // partitions inside mbgraph_run_impl right now
static ThreadBudget g_host_threads;
struct BudgetGuard {
  int n; double waited;
  explicit BudgetGuard(int k) : n(k) {
    auto t0 = std::chrono::steady_clock::now();
    g_host_threads.acquire(n);
    waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  ~BudgetGuard() { g_host_threads.release(n); }
};

// a context of its own (same device, own stream) for every graph thread: shn_thread_ctx (core.hip)

// eight 2-bit codes (one per byte, all < 4) -> their letters: 'A' + 2 b0 + 6 b1 + 11 (b0 & b1) = A, C, G, T (no carry leaves a byte)
static inline uint64_t codes8_to_ascii(uint64_t x) {
  const uint64_t b0 = x & 0x0101010101010101ULL, b1 = (x >> 1) & 0x0101010101010101ULL;
  return 0x4141414141414141ULL + 2 * b0 + 6 * b1 + 11 * (b0 & b1);
}
static void decode_read(char* s, const uint8_t* p, uint64_t n, int enc, bool rc) {
  if (enc == SHN_ENC_CODES) {
    // eight bases at a time where all eight are ACGT (a byte >= 4 anywhere in the word: the word goes base by base)
    uint64_t i = 0;
    if (!rc) {
      for (; i + 8 <= n; i += 8) {
        uint64_t x; memcpy(&x, p + i, 8);
        if (x & 0xFCFCFCFCFCFCFCFCULL) { for (uint64_t j = i; j < i + 8; j++) s[j] = p[j] < 4 ? "ACGT"[p[j]] : 'N'; continue; }
        x = codes8_to_ascii(x); memcpy(s + i, &x, 8);
      }
      for (; i < n; i++) s[i] = p[i] < 4 ? "ACGT"[p[i]] : 'N';
    } else {
      for (; i + 8 <= n; i += 8) {
        uint64_t x; memcpy(&x, p + n - 8 - i, 8);
        if (x & 0xFCFCFCFCFCFCFCFCULL) { for (uint64_t j = i; j < i + 8; j++) { const uint8_t c = p[n - 1 - j]; s[j] = c < 4 ? "TGCA"[c] : 'N'; } continue; }
        x = codes8_to_ascii(__builtin_bswap64(0x0303030303030303ULL - x)); memcpy(s + i, &x, 8);
      }
      for (; i < n; i++) { uint8_t c = p[n - 1 - i]; s[i] = c < 4 ? "TGCA"[c] : 'N'; }
    }
  } else {
    auto up = [](uint8_t c) -> char { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : (char)c; };
    if (!rc) for (uint64_t i = 0; i < n; i++) s[i] = up(p[i]);
    else for (uint64_t i = 0; i < n; i++) {
      char c = up(p[n - 1 - i]);
      s[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    }
  }
}

struct Graph {
  bool laps = false;                        // print the time of every phase (SHN_DEBUG / SHN_GRAPH_LAPS)
  shn_ctx* ctx = nullptr;                   // non-NULL: K-mer seed scans run on the GPU (csrc/seeds.hip)
  int K, L, SIZE_THRESHOLD;
  std::vector<std::string> bases;
  std::vector<std::vector<int>> ine, oute;
  std::vector<double> norm, cc, prev, cnt;
  std::vector<char> cc_int;
  std::vector<char> dead;
  std::vector<std::vector<RI>> nreads;
  std::vector<signed char> bridged;   // -1 None, 0 False, 1 True
  std::vector<int> hash;
  std::vector<int> order;
  std::vector<int> es, ed, ew;
  std::vector<double> ecc;
  // Lazy read text (reads named by rows of the host code matrices, shn_mbgraph_run_rows): nearly every read is settled on the
  // device (distinct reads, bridging seeds, the in-node test of known_paths); the text of a read is decoded from its row the first
  // time host code asks for it -- bridging hits, reads that run past their node: a few per cent of the reads.  Host code that reads
  // text from several threads calls ensure_all_text() first.
  const uint8_t *lz_a = nullptr, *lz_b = nullptr;
  uint32_t lz_L = 0;
  char* lz_buf = nullptr;
  mutable std::vector<uint64_t> lz_done;
  void decode_lazy(int r) const {
    const uint8_t* p = ((origin_flag[r] & 1) ? lz_b : lz_a) + (uint64_t)origin_row[r] * lz_L;
    decode_read(lz_buf + (size_t)r * lz_L, p, lz_L, SHN_ENC_CODES, (origin_flag[r] & 2) != 0);
    // (several threads may ask for text at once -- the X-nodes of a bridging pass: two of them decoding one read write the same
    // bytes; the bit is set after the text, with release / acquire order)
    __atomic_fetch_or(&lz_done[(size_t)r >> 6], 1ULL << (r & 63), __ATOMIC_RELEASE);
  }
  bool lz_has(size_t r) const { return (__atomic_load_n(&lz_done[r >> 6], __ATOMIC_ACQUIRE) >> (r & 63)) & 1; }
  void ensure_all_text() const {
    if (!lz_buf) return;
    const size_t n = n_rd();
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::max(1, shn_host_cpus() / 2), n >> 16));
    auto work = [&](size_t lo, size_t hi) {                       // (whole 64-read words per thread: the done bits are not shared)
      for (size_t r = lo; r < hi; r++) if (!lz_has(r)) decode_lazy((int)r);
    };
    if (nt <= 1) { work(0, n); return; }
    std::vector<std::thread> th;
    const size_t words = (n + 63) / 64;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, std::min(n, words * t / nt * 64), std::min(n, words * (t + 1) / nt * 64));
    for (auto& x : th) x.join();
  }
  RStr rstr(int r) const {
    if (lz_buf) {
      if (!lz_has((size_t)r)) decode_lazy(r);
      return RStr{lz_buf + (size_t)r * lz_L, lz_L};
    }
    return RStr{rindex.data(r), rindex.len(r)};
  }
  // the text of read r is about to be asked for: its cache lines (or, not yet decoded, the lines of its row) on their way
  void prefetch_read(int r) const {
    if (lz_buf) {
      if (!lz_has((size_t)r)) {
        const uint8_t* p = ((origin_flag[r] & 1) ? lz_b : lz_a) + (uint64_t)origin_row[r] * lz_L;
        __builtin_prefetch(p); __builtin_prefetch(p + 64);
        return;
      }
      const char* t = lz_buf + (size_t)r * lz_L;
      __builtin_prefetch(t); __builtin_prefetch(t + 64);
      return;
    }
    __builtin_prefetch(rindex.data(r));
  }
  // device-resident read attributes (graph_dev.h): copies, mates, roles, known-path states stay on the device; the host vectors
  // rcc / rmate / rmp / rfirst / rlast / rhas are then EMPTY until need_host_attrs() fetches them for a host form that wants them
  shn_dedup* dd = nullptr;
  shn_kp* kp = nullptr;
  bool dev_attrs = false;
  size_t n_rd_dev = 0;
  std::vector<uint32_t> mate_cand;                  // (a, b) node pairs of find_mate_pairs' pass over the reads, made on the device
  bool mate_cand_ready = false;
  int attrs_rc = 0;                                 // a failed fetch of the host arrays (find_known_paths is void)
  size_t n_rd() const { return dev_attrs || dd ? n_rd_dev : rindex.size(); }
  int need_host_attrs() {
    if (!dev_attrs) return 0;
    const size_t nd = n_rd_dev;
    std::vector<uint32_t> c(nd); std::vector<int32_t> m(nd); std::vector<uint8_t> ro(nd);
    int rc = shn_dedup_attrs(dd, c.data(), m.data(), ro.data());
    if (rc) return rc;
    rcc.resize(nd); rmate.resize(nd); rmp.resize(nd); rfirst.assign(nd, -1); rlast.assign(nd, -1); rhas.assign(nd, 0);
    for (size_t i = 0; i < nd; i++) { rcc[i] = (double)c[i]; rmate[i] = m[i]; rmp[i] = ro[i]; }
    dev_attrs = false;
    return 0;
  }
  void release_dev_attrs() { if (kp) { shn_kp_destroy(kp); kp = nullptr; } if (dd) { shn_dedup_destroy(dd); dd = nullptr; } }
  ~Graph() { release_dev_attrs(); }
  std::vector<double> rcc;
  std::vector<int> rmate, rmp;        // rmp: 0 None, 1, 2
  std::vector<int> rfirst, rlast;       // first / last node of the read's (last) path: all find_mate_pairs reads of Read.nodes
  std::vector<char> rhas;
  StringInterner rindex{1 << 16};
  std::set<std::vector<int>> known_paths;
  std::map<std::pair<int, int>, double> known_edges;
  std::vector<int> bridged_log;
  int n_known = 0, n_mate = 0;
  int nodes_after[4] = {0, 0, 0, 0};  // condensing, suspicious, collapse, bridging
  int final_nodes = 0;
  bool precondensed = false;          // loaded from GPU unitigs: the first condense_all is already done

  int new_node(const std::string& b) {
    int n = (int)bases.size();
    bases.push_back(b);
    ine.emplace_back(); oute.emplace_back();
    norm.push_back(1.0); cc.push_back(0.0); prev.push_back(0.0); cnt.push_back(1.0); cc_int.push_back(0);
    dead.push_back(0); nreads.emplace_back(); bridged.push_back(-1); hash.push_back(-2);
    order.push_back(n);
    return n;
  }
  void reserve_nodes(size_t n) {
    bases.reserve(n); ine.reserve(n); oute.reserve(n); norm.reserve(n); cc.reserve(n); prev.reserve(n); cnt.reserve(n);
    cc_int.reserve(n); dead.reserve(n); nreads.reserve(n); bridged.reserve(n); hash.reserve(n); order.reserve(n);
  }
  int link(int a, int b, int w) {
    int e = (int)es.size();
    es.push_back(a); ed.push_back(b); ew.push_back(w); ecc.push_back(0.0);
    oute[a].push_back(e); ine[b].push_back(e);
    return e;
  }
  static void erase_first(std::vector<int>& v, int x) {
    auto it = std::find(v.begin(), v.end(), x);
    if (it != v.end()) v.erase(it);
  }
  void kill_edge(int e) {
    erase_first(oute[es[e]], e);
    erase_first(ine[ed[e]], e);
    es[e] = -1; ed[e] = -1;
  }
  void kill_node(int n) { dead[n] = 1; }
  void full_destroy(int n) {
    std::vector<int> a = ine[n];
    for (int e : a) kill_edge(e);
    std::vector<int> b = oute[n];
    for (int e : b) kill_edge(e);
    nreads[n].clear();
    kill_node(n);
  }
  void remove_destroyed() {
    std::vector<int> o;
    o.reserve(order.size());
    for (int n : order) if (!dead[n]) o.push_back(n);
    order.swap(o);
  }
  std::vector<int> succ(int n) const { std::vector<int> r; for (int e : oute[n]) r.push_back(ed[e]); return r; }
  std::vector<int> pred(int n) const { std::vector<int> r; for (int e : ine[n]) r.push_back(es[e]); return r; }
  bool is_xnode(int n) const { return ine[n].size() >= 2 && oute[n].size() >= 2; }
  double avg_prev(int n) const { return prev[n] / cnt[n]; }

  // ---- loading (multibridging.py:145-172, 22-30, 68-97; mbgraph.py:44-62)
  void load_k1mers(const uint8_t* rows, uint64_t n_rows) {
    StringInterner idx(n_rows * 2);               // K-mer -> node (node ids == intern ids: nodes are only made here)
    idx.arena.reserve(n_rows * (size_t)K * 5 / 4);
    const int k1 = K + 1;
    reserve_nodes(n_rows * 5 / 2);
    es.reserve(n_rows * 4); ed.reserve(n_rows * 4); ew.reserve(n_rows * 4); ecc.reserve(n_rows * 4);
    for (uint64_t i = 0; i < n_rows; i++) {
      const char* km = (const char*)rows + i * k1;
      bool fresh;
      int na = idx.intern(km, K, &fresh);
      if (fresh) new_node(std::string(km, K));
      int nb = idx.intern(km + 1, K, &fresh);
      if (fresh) new_node(std::string(km + 1, K));
      link(na, nb, K - 1);
    }
    for (int n : order) { double s = 0; for (int e : oute[n]) s += ew[e]; prev[n] = s; }
  }
  // The same state as load_k1mers + the first condense_all, from the unitigs the GPU contracted (csrc/graph_gpu.hip): final
  // nodes in creation order, edges in id order, every node's out-/in-list in the order the sequential merges leave it.
  // bases / base_off: the partition's nodes; n_len = K-mers per node; tail_out = out-degree of the node's last K-mer.
  void load_unitigs(const char* ub, const uint64_t* base_off, const uint32_t* n_len, const uint32_t* tail_out, uint64_t n_nodes,
                    const uint32_t* e_src, const uint32_t* e_dst, const uint32_t* e_out_rank, const uint32_t* e_in_rank, uint64_t n_edges) {
    reserve_nodes(n_nodes * 2 + 16);
    es.reserve(n_edges * 3); ed.reserve(n_edges * 3); ew.reserve(n_edges * 3); ecc.reserve(n_edges * 3);
    for (uint64_t i = 0; i < n_nodes; i++) {
      const int n = new_node(std::string(ub + base_off[i], base_off[i + 1] - base_off[i]));
      const double m = (double)n_len[i];
      cnt[n] = m; norm[n] = m;                                       // sums of the K-mers' 1.0s
      prev[n] = (double)(K - 1) * ((m - 1.0) + (double)tail_out[i]);  // sum of (K-1) * out-degree over the K-mers (multibridging.py:169 quirk)
    }
    std::vector<uint32_t> odeg(n_nodes, 0), ideg(n_nodes, 0);
    for (uint64_t x = 0; x < n_edges; x++) { odeg[e_src[x]]++; ideg[e_dst[x]]++; }
    for (uint64_t i = 0; i < n_nodes; i++) { oute[i].assign(odeg[i], -1); ine[i].assign(ideg[i], -1); }
    for (uint64_t x = 0; x < n_edges; x++) {
      const int e = (int)es.size();
      es.push_back((int)e_src[x]); ed.push_back((int)e_dst[x]); ew.push_back(K - 1); ecc.push_back(0.0);
      oute[e_src[x]][e_out_rank[x]] = e;
      ine[e_dst[x]][e_in_rank[x]] = e;
    }
  }
  // order-sensitive signature of the graph (development check of load_unitigs against load_k1mers + condense_all)
  std::string signature() const {
    std::vector<int> idx(bases.size(), -1);
    for (size_t i = 0; i < order.size(); i++) idx[order[i]] = (int)i;
    std::string sg;
    char buf[96];
    for (int n : order) {
      sg += bases[n];
      snprintf(buf, sizeof buf, "|%.17g|%.17g|%.17g|%.17g|o", cnt[n], prev[n], norm[n], cc[n]);
      sg += buf;
      for (int e : oute[n]) { snprintf(buf, sizeof buf, " %d:%d", idx[ed[e]], ew[e]); sg += buf; }
      sg += "|i";
      for (int e : ine[n]) { snprintf(buf, sizeof buf, " %d:%d", idx[es[e]], ew[e]); sg += buf; }
      sg += "\n";
    }
    // edge ids in creation order (relative): the pairs (source, destination) of the live edges by id
    sg += "E";
    for (size_t e = 0; e < es.size(); e++) if (es[e] >= 0) { snprintf(buf, sizeof buf, " %d>%d", idx[es[e]], idx[ed[e]]); sg += buf; }
    return sg;
  }
  int add_read(const char* b, size_t n, uint64_t h) {
    bool is_new = false;
    int r = rindex.intern_hashed(b, n, h, &is_new);
    if (!is_new) { rcc[r] += 1.0; return r; }
    rcc.push_back(1.0); rmate.push_back(-1); rmp.push_back(0); rfirst.push_back(-1); rlast.push_back(-1); rhas.push_back(0);
    return r;
  }

  // ---- condensing (mbgraph.py:184-271, 479-498, 1315-1321)
  int condense(int e) {
    int s = es[e], d = ed[e], w = ew[e];
    int c = new_node(bases[s] + bases[d].substr(w));
    if (s != d) { cnt[c] = cnt[s] + cnt[d]; prev[c] = prev[s] + prev[d]; }
    else { cnt[c] = cnt[s]; prev[c] = prev[s]; }
    norm[c] = norm[s] + norm[d];
    if (norm[c] == 0) cc[c] = cc[s] + cc[d];
    else cc[c] = (cc[s] * norm[s] + cc[d] * norm[d]) / norm[c];
    if (s == d) {
      kill_edge(e);
      { std::vector<int> t = oute[s]; for (int x : t) { int ne = link(c, ed[x], ew[x]); ecc[ne] = ecc[x]; kill_edge(x); } }
      { std::vector<int> t = ine[s]; for (int x : t) { int ne = link(es[x], c, ew[x]); ecc[ne] = ecc[x]; kill_edge(x); } }
      cc[c] = cc[s] / 2.0;
      norm[c] = norm[s];
      nreads[c] = nreads[s];
      nreads[s].clear();
      kill_node(s);
      return c;
    }
    { std::vector<int> t = ine[s]; for (int x : t) { link(es[x], c, ew[x]); kill_edge(x); } }
    { std::vector<int> t = oute[d]; for (int x : t) { link(c, ed[x], ew[x]); kill_edge(x); } }
    int shift = (int)bases[s].size() - w;
    // P1: the source's reads sorted by (read, index), without repeats; then the destination's that are not among them (a sorted
    // vector + bisection: a tree node per read was most of bridge_all for the X-nodes of a highly expressed transcript)
    const double tcs0 = laps ? tnow() : 0.0;
    std::vector<RI> src;
    src.swap(nreads[s]);                                             // (the source's list is given up below anyway)
    sort_runs(src);
    // the source's reads without repeats, then the destination's that are not among them, in ONE vector of the final size (the
    // lists of a highly expressed transcript hold 10^5 reads: no second copy when the destination's part is appended)
    std::vector<RI> out;
    out.reserve(src.size() + nreads[d].size());
    for (size_t q = 0; q < src.size(); q++) if (q == 0 || !(src[q] == src[q - 1])) out.push_back(src[q]);
    const size_t n_src = out.size();
    if (std::is_sorted(nreads[d].begin(), nreads[d].end())) {
      // (both lists ascending -- the shift keeps the order: one pass with a cursor into the source's part instead of a bisection per read)
      size_t a = 0;
      for (const RI& x : nreads[d]) {
        const RI y(x.first, x.second - shift);
        while (a < n_src && out[a] < y) a++;
        if (!(a < n_src && out[a] == y)) out.push_back(y);
      }
    } else
      for (const RI& x : nreads[d]) { RI y(x.first, x.second - shift); if (!std::binary_search(out.begin(), out.begin() + (ptrdiff_t)n_src, y)) out.push_back(y); }
    nreads[c].swap(out);
    if (laps) { t_cond_sort += tnow() - tcs0; v_cond += nreads[c].size(); }
    nreads[s].clear(); nreads[d].clear();
    kill_edge(e);
    kill_node(s); kill_node(d);
    return c;
  }
  void local_condense_edge(int e) {
    if (es[e] < 0) return;
    if (oute[es[e]].size() > 1 || ine[ed[e]].size() > 1) return;
    int c = condense(e);
    std::vector<int> t = ine[c];
    t.insert(t.end(), oute[c].begin(), oute[c].end());
    for (int x : t) local_condense_edge(x);
  }
  void local_condense_node(int n) {
    if (oute[n].size() == 1) local_condense_edge(oute[n][0]);
    if (ine[n].size() == 1) local_condense_edge(ine[n][0]);
  }
  void condense_all() {
    size_t i = 0;
    while (i < order.size()) {
      int n = order[i++];
      if (oute[n].size() != 1) continue;
      int e = oute[n][0], d = ed[e];
      if (ine[d].size() == 1 && n != d) condense(e);
    }
    remove_destroyed();
  }

  // ---- error pruning (mbgraph.py:1169-1313)
  bool is_suspicious(int n) const {
    size_t ni = ine[n].size(), no = oute[n].size();
    if ((int)bases[n].size() <= SIZE_THRESHOLD && (ni == 0 || no == 0)) return true;
    if (avg_prev(n) >= 1) return false;
    if (ni == 0 || no == 0) return true;
    { double t = 0; for (int e : ine[n]) t += (double)oute[es[e]].size(); if (t / (double)ni < 2) return false; }
    { double t = 0; for (int e : oute[n]) t += (double)ine[ed[e]].size(); if (t / (double)no < 2) return false; }
    return true;
  }
  void destroy_suspicious() {
    while (true) {
      std::vector<int> sus;
      for (int n : order) if (is_suspicious(n)) sus.push_back(n);
      if (sus.empty()) return;
      std::stable_sort(sus.begin(), sus.end(), [&](int a, int b) { return avg_prev(a) < avg_prev(b); });
      for (int n : sus) {
        if (dead[n]) continue;
        std::vector<int> adj = pred(n), s2 = succ(n);
        adj.insert(adj.end(), s2.begin(), s2.end());
        full_destroy(n);
        for (int a : adj) local_condense_node(a);
      }
      remove_destroyed();
    }
  }
  bool similar(int a, int b) const {
    if (dead[a] || dead[b]) return false;
    const std::string &x = bases[a], &y = bases[b];
    if (x.size() != y.size()) return false;
    size_t mism = 0;
    for (size_t i = 0; i < x.size(); i++) mism += x[i] != y[i];
    if ((double)mism / (double)std::max<size_t>(x.size(), 1) >= 0.1) return false;
    auto uniq = [](std::vector<int> v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); return v; };
    return uniq(succ(a)) == uniq(succ(b)) && uniq(pred(a)) == uniq(pred(b));
  }
  void collapse_all() {
    while (true) {
      bool collapsed = false;
      for (size_t oi = 0; oi < order.size(); oi++) {
        int n = order[oi];
        std::vector<int> ss;
        for (int s : succ(n)) if (!dead[s]) ss.push_back(s);
        bool done = false;
        for (size_t i = 0; i < ss.size() && !done; i++)
          for (size_t j = i + 1; j < ss.size(); j++)
            if (similar(ss[i], ss[j])) {
              int a = ss[i], b = ss[j];
              if (prev[a] < prev[b]) std::swap(a, b);
              prev[a] += prev[b];
              full_destroy(b);
              collapsed = done = true;
              break;
            }
      }
      remove_destroyed();
      if (!collapsed) return;
    }
  }

  // ---- bridging (mbgraph.py:77-111, 450-628)
  size_t rlen(int r) const { return lz_buf ? (size_t)lz_L : rindex.len(r); }
  // a hit of the device's seed scan (the read holds the node's first K-mer at `index`, bit for bit): for a node that IS one K-mer --
  // nearly every X-node -- the text is matched already and what is left of read_bridges is its bounds, without a look at the read
  bool hit_bridges(int r, int n, int index) const {
    if ((int)bases[n].size() == K) return index > 0 && (int)rlen(r) > index + K;
    return read_bridges(r, n, index);
  }
  bool read_bridges(int r, int n, int index) const {
    const RStr rb = rstr(r); const std::string& nb = bases[n];
    if (index <= 0 || (int)rb.size() <= index + (int)nb.size()) return false;
    return rb.compare(index, nb.size(), nb) == 0;
  }
  // packed key of s[pos..pos+K) or false if it holds a non-ACGT character
  template <class S> bool key_at(const S& s, size_t pos, uint64_t& key) const {
    key = 0;
    for (int j = 0; j < K; j++) { int c = base_code(s[pos + j]); if (c < 0) return false; key = (key << 2) | (uint64_t)c; }
    return true;
  }
  int acgt_known = -1;        // set while the reads are loaded (a scan of the arena is 10s of ms at the read cap); -1 = scan
  bool reads_all_acgt() {
    if (acgt_known < 0) {
      acgt_known = 1;
      for (char c : rindex.arena) if (base_code(c) < 0) { acgt_known = 0; break; }
    }
    return acgt_known == 1;
  }
  // device copy of the distinct reads + pattern table; returns false if the GPU path is not usable
  // the distinct reads on the device: rows of the resident input (device gather; 5 bytes per read cross the bus), else their text
  bool ensure_dreads(shn_reads** dreads) {
    if (!ctx || n_rd() == 0 || !reads_all_acgt()) return false;
    if (*dreads) return true;
    if (src_a && dd && dd->n_distinct == n_rd()) return shn_reads_gather_dev(ctx, src_a, src_b, dd->d_row, dd->d_flag, n_rd(), dreads) == 0;
    if (src_a && origin_row.size() == n_rd()) return shn_reads_gather(ctx, src_a, src_b, origin_row.data(), origin_flag.data(), n_rd(), dreads) == 0;
    return shn_reads_create(ctx, (const uint8_t*)rindex.arena.data(), rindex.off.data(), n_rd(), 0, SHN_ENC_ASCII, dreads) == 0;
  }
  bool gpu_patterns(const SeedIndex& si, shn_reads** dreads, shn_table** tab) {
    if (!ctx || K > 32 || n_rd() == 0 || si.keys.empty() || !reads_all_acgt()) return false;
    if (!ensure_dreads(dreads)) return false;
    std::vector<uint32_t> vals(si.keys.size());
    for (size_t i = 0; i < vals.size(); i++) vals[i] = (uint32_t)i + 1;
    return shn_table_create(ctx, si.keys.data(), vals.data(), si.keys.size(), K, 0, tab) == 0;
  }
  shn_reads* d_reads = nullptr;
  const shn_reads *src_a = nullptr, *src_b = nullptr;      // the resident input read sets the partition's reads are rows of
  std::vector<uint32_t> origin_row;                        // per distinct read: row in its set, and
  std::vector<uint8_t> origin_flag;                        // bit 0: set b, bit 1: reverse complement
  void release_gpu() { if (d_reads) { shn_reads_destroy(d_reads); d_reads = nullptr; } }

  void find_bridging_reads() {
    std::vector<std::pair<uint64_t, std::pair<int, int>>> items;
    for (int n : order) if (is_xnode(n)) { uint64_t key; if (key_at(bases[n], 0, key)) items.push_back({key, {n, 0}}); }
    if (items.empty()) return;
    SeedIndex si;
    si.build(items);
    shn_table* tab = nullptr;
    {
      // GPU section (on this thread's own context / stream)
      bool gpu_ok = false;
      uint64_t nh = 0;
      std::vector<uint32_t> hr, hs, hi;
      {
        const double tf0 = laps ? tnow() : 0.0;
        double tf1 = tf0, tf2 = tf0;
        if (gpu_patterns(si, &d_reads, &tab)) {
          if (laps) tf1 = tnow();
          gpu_ok = shn_seed_scan(ctx, d_reads, K, tab, &nh, nullptr, nullptr, nullptr) == 0;
          if (laps) tf2 = tnow();
          if (gpu_ok && nh) {
            hr.resize(nh); hs.resize(nh); hi.resize(nh);
            gpu_ok = shn_seed_scan(ctx, d_reads, K, tab, &nh, hr.data(), hs.data(), hi.data()) == 0;
          }
          shn_table_destroy(tab);
        }
        if (laps) fprintf(stderr, "[mbgraph]   find_bridging_reads: reads + patterns on the device %.3f s, count pass %.3f s, fill pass + download %.3f s (%llu hits, %zu patterns)\n",
                          tf1 - tf0, tf2 - tf1, tnow() - tf2, (unsigned long long)nh, si.keys.size());
      }
      if (gpu_ok) {
        // the hits against the node texts (read_bridges: the read spells the whole node with a base to spare on either side) -- on the
        // host threads that are free right now when there are many (the largest partition of BASELINE configs[2]: 415 k hits, 30-58 ms
        // on one thread, with the other partitions long done): slices of the hit list in order, each thread's confirmed hits kept in
        // order and appended slice by slice, so every node's list is what the one-thread loop makes
        const unsigned want = nh >= 100000 ? (unsigned)std::min<uint64_t>(8, nh / 50000) : 1u;
        const unsigned nt = want > 1 ? (unsigned)g_host_threads.take_free((int)want, g_partitions_running.load() - 1) : 1u;
        struct GiveBackH { unsigned n; ~GiveBackH() { if (n) g_host_threads.release((int)n); } } give_back_h{want > 1 ? nt : 0u};
        if (nt > 1) {
          std::vector<std::vector<std::pair<int, RI>>> found(nt);
          auto slice = [&](unsigned t) {
            std::vector<std::pair<int, RI>>& out = found[t];
            const uint64_t h0 = nh * t / nt, h1 = nh * (t + 1) / nt;
            for (uint64_t h = h0; h < h1; h++) {
              if (h + 12 < h1) prefetch_read((int)hr[h + 12]);
              for (uint32_t q = si.goff[hi[h]]; q < si.goff[hi[h] + 1]; q++) {
                const int x = si.occ[q].first;
                if (hit_bridges((int)hr[h], x, (int)hs[h])) out.push_back({x, RI((int)hr[h], (int)hs[h])});
              }
            }
          };
          std::vector<std::thread> th;
          for (unsigned t = 1; t < nt; t++) th.emplace_back(slice, t);
          slice(0);
          for (auto& x : th) x.join();
          for (unsigned t = 0; t < nt; t++) for (const auto& xr : found[t]) nreads[xr.first].push_back(xr.second);
          return;
        }
        for (uint64_t h = 0; h < nh; h++) {
          if (h + 12 < nh) prefetch_read((int)hr[h + 12]);
          for (uint32_t q = si.goff[hi[h]]; q < si.goff[hi[h] + 1]; q++) {
            int x = si.occ[q].first;
            if (hit_bridges((int)hr[h], x, (int)hs[h])) nreads[x].push_back(RI((int)hr[h], (int)hs[h]));
          }
        }
        return;
      }
    }
    const uint64_t mask = K == 32 ? ~0ULL : ((1ULL << (2 * K)) - 1);
    for (int r = 0; r < (int)n_rd(); r++) {
      const RStr rb = rstr(r);
      uint64_t key = 0;
      int valid = 0;
      for (int i = 0; i < (int)rb.size(); i++) {
        int c = base_code(rb[i]);
        if (c < 0) { valid = 0; key = 0; continue; }
        key = ((key << 2) | (uint64_t)c) & mask;
        if (++valid < K) continue;
        int s0 = i - K + 1;                                  // window start; the reference scans range(1, len-K)
        if (s0 < 1 || s0 >= (int)rb.size() - K) continue;
        int gi = si.find(key);
        if (gi < 0) continue;
        for (uint32_t q = si.goff[gi]; q < si.goff[gi + 1]; q++) {
          int x = si.occ[q].first;
          if (read_bridges(r, x, s0)) nreads[x].push_back(RI(r, s0));
        }
      }
    }
  }
  void refresh_bridging_reads(int n) {
    const double tr0 = laps ? tnow() : 0.0;
    struct Lap { Graph* g; double t0; size_t v; ~Lap() { if (g->laps) { g->t_refresh += tnow() - t0; g->n_refresh++; g->v_refresh += v; } } } lap_{this, tr0, nreads[n].size()};
    const std::string& nb = bases[n];
    int lb = (int)nb.size();
    // (the reference collects them in a set and the pinned order P1 walks it sorted: a sorted vector without repeats is the same
    // sequence -- the lists of a highly expressed X-node hold 10^5 reads, and a tree insert per read was most of bridge_all)
    // the characters the neighbours put in front of / behind the node (the same for every read of the list)
    bool in_ch[256] = {false}, out_ch[256] = {false};
    if (!nreads[n].empty()) {
      for (int e : ine[n]) { const std::string& pb = bases[es[e]]; in_ch[(unsigned char)pb[pb.size() - ew[e] - 1]] = true; }
      for (int e : oute[n]) out_ch[(unsigned char)bases[ed[e]][ew[e]]] = true;
    }
    // one pass over the list: the read spells the node from i on, and the characters before and behind are ones a neighbour offers
    // (the reference filters by text, orders, then filters by the neighbours: the filters commute, the order comes last here)
    std::vector<RI> rs;
    rs.reserve(nreads[n].size());
    const std::vector<RI>& in_list = nreads[n];
    for (size_t q = 0; q < in_list.size(); q++) {
      if (q + 8 < in_list.size()) prefetch_read(in_list[q + 8].first);
      const RI& x = in_list[q];
      const int r = x.first, i = x.second;
      if (i <= 0) continue;
      const RStr rb = rstr(r);
      if ((int)rb.size() > i + lb && in_ch[(unsigned char)rb[i - 1]] && out_ch[(unsigned char)rb[i + lb]] && rb.compare(i, lb, nb) == 0) rs.push_back(x);
    }
    sort_runs(rs);
    rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
    nreads[n].swap(rs);
  }
  // bridge_all asks every X-node in every pass; the answer depends on the node's edge lists (edge ids: an edge never changes its
  // ends or its weight), on its bridging reads (which only this function filters, idempotently, between two passes -- bridging
  // appends reads to the NEW nodes it makes) and on texts that never change: an X-node whose edge lists and read count are what
  // they were at the last call gets the last answer, without going through its 10^3-10^5 reads again
  struct BridgedMemo { std::vector<int> in, out; size_t n_reads = (size_t)-1; bool answer = false; };
  std::vector<BridgedMemo> bridged_memo;
  bool is_bridged_xnode(int n) {
    if (bridged_memo.size() < bases.size()) bridged_memo.resize(bases.size());
    BridgedMemo& memo = bridged_memo[n];
    if (memo.n_reads == nreads[n].size() && memo.in == ine[n] && memo.out == oute[n]) return memo.answer;
    const bool ans = is_bridged_xnode_now(n);
    BridgedMemo& m2 = bridged_memo[n];
    m2.in = ine[n]; m2.out = oute[n]; m2.n_reads = nreads[n].size(); m2.answer = ans;
    return ans;
  }
  bool is_bridged_xnode_now(int n) {
    refresh_bridging_reads(n);
    int lb = (int)bases[n].size();
    bool inb[256] = {false}, outb[256] = {false};                  // (the distinct characters before / behind the node in its reads)
    int n_in = 0, n_out = 0;
    for (const RI& x : nreads[n]) {
      const RStr rb = rstr(x.first);
      const unsigned char a = (unsigned char)rb[x.second - 1], b = (unsigned char)rb[x.second + lb];
      if (!inb[a]) { inb[a] = true; n_in++; }
      if (!outb[b]) { outb[b] = true; n_out++; }
    }
    int bi = (int)ine[n].size() - n_in, bo = (int)oute[n].size() - n_out;
    return (bi == 0 && bo == 0) || (bi == 1 && bo == 1);
  }
  int bridging_step(int node) {
    // (bridge_all has just asked is_bridged_xnode about this node: unless a bridging step in between touched its edges, its reads
    // are filtered already)
    if (!((size_t)node < bridged_memo.size() && bridged_memo[node].n_reads == nreads[node].size() && bridged_memo[node].in == ine[node] &&
          bridged_memo[node].out == oute[node]))
      refresh_bridging_reads(node);
    if (nreads[node].empty() || ine[node].size() < 2 || oute[node].size() < 2) return shn_fail(SHN_ERR_INTERNAL, "bridging_step: assertion failed");
    const std::string nb = bases[node];
    int lb = (int)nb.size();
    std::vector<int> u_list, w_list;
    int v_back = -1, v_forward = -1, loop_w = 0;
    double tb0 = laps ? tnow() : 0.0;
    if (laps) { n_steps++; v_bs += nreads[node].size(); }
    { std::vector<int> t = ine[node];
      for (int e : t) {
        int p = es[e], w = ew[e];
        const std::string& pb = bases[p];
        const char cp = pb[pb.size() - w - 1];
        int u = new_node(std::string(1, cp) + nb);
        link(p, u, w + 1);
        bridged[u] = 0;
        if (p == node) { v_back = u; loop_w = w; }
        u_list.push_back(u);
      }
      // read_bridges(read, u, i - 1) for a read the refresh above left in the list (i > 0, the read longer than i + lb, its text equal
      // to the node's from i on): the read reaches one base further back and that base is u's first -- no text to compare again.
      // One pass over the reads for all the u-nodes (each gets the reads with its character, in list order).
      const std::vector<RI>& lst = nreads[node];
      for (size_t q = 0; q < lst.size(); q++) {
        if (q + 8 < lst.size()) prefetch_read(lst[q + 8].first);
        const RI& x = lst[q];
        if (x.second - 1 <= 0) continue;
        const char c = rstr(x.first)[x.second - 1];
        for (int u : u_list) if (bases[u][0] == c) nreads[u].push_back(RI(x.first, x.second - 1));
      } }
    if (laps) { const double t = tnow(); t_bs_in += t - tb0; tb0 = t; }
    { std::vector<int> t = oute[node];
      for (int e : t) {
        int q = ed[e], w = ew[e];
        int x = new_node(nb + std::string(1, bases[q][w]));
        link(x, q, w + 1);
        for (const RI& y : nreads[node]) if (read_bridges(y.first, x, y.second + 1)) nreads[x].push_back(RI(y.first, y.second + 1));
        bridged[x] = 0;
        if (q == node) v_forward = x;
        w_list.push_back(x);
      } }
    if (laps) { const double t = tnow(); t_bs_out += t - tb0; tb0 = t; }
    { std::vector<int> t = ine[node]; for (int e : t) kill_edge(e); }
    { std::vector<int> t = oute[node]; for (int e : t) kill_edge(e); }
    if (v_back >= 0) {
      if (v_forward < 0) return shn_fail(SHN_ERR_INTERNAL, "bridging_step: self loop without forward node");
      link(v_forward, v_back, loop_w + 2);
    }
    std::map<int, int> links;
    for (int n : u_list) links[n] = 0;
    for (int n : w_list) links[n] = 0;
    std::vector<RI> rl = nreads[node];
    // (the u-nodes are `base + node text`, the w-nodes `node text + base` -- all lb + 1 bases long --, and the read equals the node
    // text from i on: one base decides; the bases looked up once, not per read)
    std::vector<char> u_ch(u_list.size()), w_ch(w_list.size());
    for (size_t a = 0; a < u_list.size(); a++) u_ch[a] = (int)bases[u_list[a]].size() == lb + 1 ? bases[u_list[a]][0] : '\0';
    for (size_t a = 0; a < w_list.size(); a++) w_ch[a] = (int)bases[w_list[a]].size() == lb + 1 ? bases[w_list[a]][lb] : '\0';
    for (size_t q = 0; q < rl.size(); q++) {
      if (q + 8 < rl.size()) prefetch_read(rl[q + 8].first);
      const RI& y = rl[q];
      const RStr rb = rstr(y.first);
      int i = y.second;
      // exactly one u-node spelling the read from i - 1 and one w-node spelling it from i
      int u = -1, x = -1, nu = 0, nw = 0;
      if (rb.size() >= (size_t)(i + lb)) { const char c = rb[i - 1]; for (size_t a = 0; a < u_ch.size(); a++) if (u_ch[a] == c && c) { u = u_list[a]; nu++; } }
      if (rb.size() >= (size_t)(i + lb + 1)) { const char c = rb[i + lb]; for (size_t a = 0; a < w_ch.size(); a++) if (w_ch[a] == c && c) { x = w_list[a]; nw++; } }
      if (nu != 1 || nw != 1) continue;
      nreads[u].push_back(RI(y.first, i - 1));
      nreads[x].push_back(RI(y.first, i));
      bool prec = false;
      for (int e : oute[u]) if (ed[e] == x) prec = true;
      if (!prec) { link(u, x, lb); bridged[u] = 1; bridged[x] = 1; links[u]++; links[x]++; }
    }
    nreads[node].clear();
    if (laps) { const double t = tnow(); t_bs_rl += t - tb0; tb0 = t; }
    std::vector<int> ub_u, ub_w;
    for (int u : u_list) if (bridged[u] != 1) ub_u.push_back(u);
    for (int x : w_list) if (bridged[x] != 1) ub_w.push_back(x);
    if (ub_u.size() == 1 && ub_w.size() == 1) { link(ub_u[0], ub_w[0], lb); links[ub_u[0]]++; links[ub_w[0]]++; }
    else if (ub_u.size() + ub_w.size() != 0) return shn_fail(SHN_ERR_INTERNAL, "bridging_step: unbridged edges remain");
    int link_count = 0;
    for (auto& kv : links) link_count += kv.second;
    std::vector<int> all = u_list;
    all.insert(all.end(), w_list.begin(), w_list.end());
    for (int n : all) prev[n] = ((double)links[n] / (double)link_count) * prev[node];
    const double tc0 = laps ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
    for (int n : all) {
      std::vector<int> t = ine[n];
      t.insert(t.end(), oute[n].begin(), oute[n].end());
      for (int e : t) local_condense_edge(e);
    }
    if (laps) t_condense += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tc0;
    kill_node(node);
    return 0;
  }
  double t_condense = 0, t_refresh = 0, t_bs_in = 0, t_bs_out = 0, t_bs_rl = 0, t_cond_sort = 0;
  size_t n_refresh = 0, v_refresh = 0, v_bs = 0, v_cond = 0, n_steps = 0;
  static double tnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  int bridge_all() {
    auto nowb = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_ask = 0, t_step = 0;
    size_t n_ask = 0;
    while (true) {
      std::vector<int> todo;
      double t0 = nowb();
      {
        // the X-nodes whose last answer does not stand any more: each one's question touches its own read list and memo only (and
        // read text, see decode_lazy), so they are asked on the host threads that are free right now -- late in the stage, when the
        // largest partition is at its bridging alone, that is most of them
        std::vector<int> stale;
        if (bridged_memo.size() < bases.size()) bridged_memo.resize(bases.size());
        for (int n : order) if (is_xnode(n)) {
          const BridgedMemo& memo = bridged_memo[n];
          if (!(memo.n_reads == nreads[n].size() && memo.in == ine[n] && memo.out == oute[n])) stale.push_back(n);
        }
        size_t work = 0;
        for (int n : stale) work += nreads[n].size();
        const unsigned want = work >= 50000 && stale.size() >= 8 ? (unsigned)std::min<size_t>(8, stale.size() / 4) : 1u;
        const unsigned nt = want > 1 ? (unsigned)g_host_threads.take_free((int)want, g_partitions_running.load() - 1) : 1u;
        struct GiveBackA { unsigned n; ~GiveBackA() { if (n) g_host_threads.release((int)n); } } give_back_a{want > 1 ? nt : 0u};
        if (nt > 1) {
          std::atomic<size_t> next{0};
          auto ask = [&]() { for (size_t i; (i = next.fetch_add(1)) < stale.size();) (void)is_bridged_xnode(stale[i]); };
          std::vector<std::thread> th;
          for (unsigned t = 1; t < nt; t++) th.emplace_back(ask);
          ask();
          for (auto& x : th) x.join();
        }
      }
      for (int n : order) if (is_xnode(n)) { n_ask++; if (is_bridged_xnode(n)) todo.push_back(n); }
      double t1 = nowb();
      for (int n : todo) { int rc = bridging_step(n); if (rc) return rc; }
      t_ask += t1 - t0; t_step += nowb() - t1;
      bridged_log.push_back((int)todo.size());
      remove_destroyed();
      if (todo.empty()) {
        if (laps) fprintf(stderr, "[mbgraph]   bridge_all: %zu passes, is_bridged_xnode %.3f s (%zu questions), bridging_step %.3f s (%.3f s of it condensing)\n",
                          bridged_log.size(), t_ask, n_ask, t_step, t_condense);
        if (laps) fprintf(stderr, "[mbgraph]   bridge_all: refresh %.3f s (%zu calls, %zu reads), steps %zu (%zu reads): in %.3f out %.3f distribute %.3f s; condense read lists %.3f s (%zu reads)\n",
                          t_refresh, n_refresh, v_refresh, n_steps, v_bs, t_bs_in, t_bs_out, t_bs_rl, t_cond_sort, v_cond);
        return 0;
      }
    }
  }

  // ---- copy counts, cycles (mbgraph.py:735-767, 903-960, 1040-1049, 1133-1161, 1324-1331)
  void find_approximate_copy_counts() {
    known_paths.clear();
    for (int n : order) { norm[n] = (double)((int)bases[n].size() - K + 1); cc[n] = prev[n] / norm[n]; cc_int[n] = 0; }
    for (int n : order)
      for (int e : oute[n]) {
        int nm = std::max(L - ew[e] - 1, 0);
        int a = es[e], b = ed[e];
        double tot = cc[a] * norm[a] + cc[b] * norm[b];
        ecc[e] = nm == 0 ? 0.0 : 0.5 * tot / (double)nm;
      }
  }
  void disregard_loops() {
    for (int n : order) { bool self = false; for (int e : oute[n]) if (ed[e] == n) self = true; if (self) { norm[n] = 0; cc[n] = 0; } }
  }
  bool reachable_cycle(int n, std::set<int>& no_cycles, std::vector<int>& trav, std::vector<int>& out) {
    trav.push_back(n);
    std::vector<int> ss = succ(n);
    for (int m : ss) {
      auto it = std::find(trav.begin(), trav.end(), m);
      if (it != trav.end()) { out.assign(it, trav.end()); out.push_back(m); trav.pop_back(); return true; }
      if (no_cycles.count(m)) continue;
      if (reachable_cycle(m, no_cycles, trav, out)) { trav.pop_back(); return true; }
    }
    no_cycles.insert(n);
    trav.pop_back();
    return false;
  }
  bool find_cycle(std::set<int>& no_cycles, std::vector<int>& out) {
    for (int n : order) {
      if (no_cycles.count(n)) continue;
      std::vector<int> trav;
      if (reachable_cycle(n, no_cycles, trav, out)) return true;
    }
    return false;
  }
  int break_cycles() {
    std::set<int> no_cycles;
    std::vector<int> c;
    while (find_cycle(no_cycles, c)) full_destroy(c[1]);
    remove_destroyed();
    condense_all();
    std::set<int> chk;
    if (find_cycle(chk, c)) return shn_fail(SHN_ERR_INTERNAL, "break_cycles: graph still cyclic");
    return 0;
  }

  // ---- reads on the graph (mbgraph.py:1355-1441, 114-160, 839-880)
  template <class S> static bool compare(const S& a, size_t ao, const std::string& b, size_t bo) {
    size_t n = std::min(a.size() - ao, b.size() - bo);
    return memcmp(&a[0] + ao, b.data() + bo, n) == 0;
  }
  template <class S> void search_sequence(const S& seq, size_t so, int node, int i, int hops, std::vector<int>& cur, std::vector<std::vector<int>>& out) {
    size_t nl = bases[node].size() - i;
    cur.push_back(node);
    if (hops <= 0 || seq.size() - so <= nl) { out.push_back(cur); cur.pop_back(); return; }
    size_t so2 = so + nl;
    for (int e : oute[node])
      if (compare(seq, so2, bases[ed[e]], ew[e])) search_sequence(seq, so2, ed[e], ew[e], hops - 1, cur, out);
    cur.pop_back();
  }
  void find_known_paths() {
    known_paths.clear();
    const bool dbgk = laps;
    auto nowk = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tk0 = nowk(), tk1 = 0, tk2 = 0, tk3 = 0;
    const uint64_t mask = K == 32 ? ~0ULL : ((1ULL << (2 * K)) - 1);
    // ---- device path (kpaths_gpu.hip): every read is classified where the reads are; the host searches only the reads that run
    // past the end of the node they start in, against an index of just their first K-mers
    { const char* kv = getenv("SHN_GRAPH_KP_GPU");
      if (ctx && K <= 31 && !(kv && kv[0] == '0') && n_rd() && ensure_dreads(&d_reads)) {
        std::string nb;
        std::vector<uint64_t> noff(1, 0);
        { size_t tot = 0; for (int n : order) tot += bases[n].size(); nb.reserve(tot); noff.reserve(order.size() + 1); }
        for (int n : order) { nb += bases[n]; noff.push_back(nb.size()); }
        std::vector<uint8_t> st(n_rd());
        std::vector<int32_t> nd(n_rd());
        std::vector<uint32_t> no(n_rd());
        // the out-edges of the nodes in list order, for the device's search of the reads that run past their first node (kp_search):
        // destination as an index into `order`, and the offset into the destination at which it continues the source
        std::vector<uint32_t> eoff(order.size() + 1, 0), edst, eov;
        {
          std::unordered_map<int, uint32_t> pos_of;
          pos_of.reserve(order.size() * 2);
          for (size_t q = 0; q < order.size(); q++) pos_of[order[q]] = (uint32_t)q;
          for (size_t q = 0; q < order.size(); q++) {
            for (int e : oute[order[q]]) { edst.push_back(pos_of[ed[e]]); eov.push_back((uint32_t)ew[e]); }
            eoff[q + 1] = (uint32_t)edst.size();
          }
        }
        // room for the records [read, length, nodes ...] of the searched reads: a few words per read that runs on (4 % of the reads do)
        std::vector<int32_t> precs(std::max<size_t>(1u << 16, n_rd() / 2 + 4096));
        uint64_t precs_used = 0;
        tk1 = nowk();
        if (dev_attrs) {
          // everything per read stays on the device: the host gets the reads left to search, the records of the reads searched
          // there and (below, after its own searches) the node pairs of find_mate_pairs
          std::vector<KpSlow> left;
          std::vector<int32_t> recs;
          std::vector<uint32_t> rec_cnt;
          if (kp) { shn_kp_destroy(kp); kp = nullptr; }
          const int rcf = shn_known_paths_dev(ctx, d_reads, K, (const uint8_t*)nb.data(), noff.data(), order.size(), eoff.data(), edst.data(), eov.data(), dd, &left,
                                              &recs, &rec_cnt, &kp);
          release_gpu();
          tk2 = nowk();
          if (rcf == 0) {
            std::unordered_map<int, int> pos_in_order;
            FlatSum cnt_of(rec_cnt.size() / 2 + 16);                 // (a searched read's copies: every read has one entry)
            for (size_t i = 0; i + 1 < rec_cnt.size(); i += 2) cnt_of.set(rec_cnt[i], (double)rec_cnt[i + 1]);
            int cntp = 0;
            FlatSum edge_sum(4096);
            std::unordered_map<uint64_t, std::vector<std::vector<int>>> seen;
            auto add_path = [&](const std::vector<int>& pth, double copies) {
              for (size_t j = 0; j + 1 < pth.size(); j++) edge_sum.add(((uint64_t)(uint32_t)pth[j] << 32) | (uint32_t)pth[j + 1], copies);
              if (pth.size() > 2) {
                cntp++;
                uint64_t h = 0xcbf29ce484222325ULL;
                for (int v : pth) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001b3ULL;
                auto& lst = seen[h];
                bool have = false;
                for (const auto& q : lst) if (q == pth) { have = true; break; }
                if (!have) { lst.push_back(pth); known_paths.insert(pth); }
              }
            };
            std::vector<int> pth;
            for (uint64_t at = 0; at + 2 <= recs.size();) {
              const int32_t r = recs[at], len = recs[at + 1];
              if (len <= 0 || at + 2 + (uint64_t)len > recs.size()) break;
              pth.resize((size_t)len);
              for (int32_t j = 0; j < len; j++) pth[(size_t)j] = order[(size_t)recs[at + 2 + (uint64_t)j]];
              at += 2 + (uint64_t)len;
              add_path(pth, cnt_of.get((uint64_t)(uint32_t)r));
            }
            tk3 = nowk();
            // the reads the device left (a search deeper than its stack, no room for the records): the host's search, one by one
            std::vector<int32_t> patches;
            if (!left.empty()) {
              for (size_t q = 0; q < order.size(); q++) pos_in_order[order[q]] = (int)q;
              std::sort(left.begin(), left.end(), [](const KpSlow& a, const KpSlow& b) { return a.read < b.read; });
              // occurrences (node, offset), in index order, of the first K-mers of the reads of state 2
              std::unordered_map<uint64_t, std::vector<std::pair<int, int>>> occ_of;
              for (const KpSlow& sl : left) if (sl.state == 2) { uint64_t key; if (key_at(rstr((int)sl.read), 0, key)) occ_of[key]; }
              if (!occ_of.empty())
                for (int n : order) {
                  const std::string& b = bases[n];
                  uint64_t key = 0;
                  for (int i = 0; i < (int)b.size(); i++) {
                    key = ((key << 2) | (uint64_t)base_code(b[i])) & mask;
                    if (i + 1 < K) continue;
                    auto it = occ_of.find(key);
                    if (it != occ_of.end()) it->second.push_back({n, i - K + 1});
                  }
                }
              std::vector<std::vector<int>> paths;
              std::vector<int> cur;
              std::vector<std::pair<int, int>> one(1);
              static const std::vector<std::pair<int, int>> none;
              for (const KpSlow& sl : left) {
                const int r = (int)sl.read;
                const RStr rb = rstr(r);
                uint64_t key;
                if (!key_at(rb, 0, key)) continue;
                const std::vector<std::pair<int, int>>* occs = &one;
                if (sl.state == 3) one[0] = {order[sl.node], (int)sl.offset};
                else { auto it = occ_of.find(key); occs = it == occ_of.end() ? &none : &it->second; }
                int pf = -1, pl = -1;
                for (const auto& oc : *occs) {
                  const int sn = oc.first, so = oc.second;
                  if (!compare(rb, 0, bases[sn], so)) continue;
                  if (rb.size() <= bases[sn].size() - (size_t)so) { pf = sn; pl = sn; continue; }
                  paths.clear(); cur.clear();
                  search_sequence(rb, 0, sn, so, 30, cur, paths);
                  for (auto& p : paths) { pf = p.front(); pl = p.back(); add_path(p, (double)sl.cnt); }
                }
                patches.push_back(r); patches.push_back(pf < 0 ? -1 : pos_in_order[pf]); patches.push_back(pl < 0 ? -1 : pos_in_order[pl]);
              }
            }
            edge_sum.each([&](uint64_t k, double v) { known_edges[{(int)(uint32_t)(k >> 32), (int)(uint32_t)k}] += v; });
            n_known = cntp;
            // find_mate_pairs' pass over the reads, where their first / last nodes are (the graph does not change in between)
            mate_cand.clear();
            mate_cand_ready = shn_kp_mate_pairs(kp, dd, patches.data(), patches.size() / 3, &mate_cand) == 0;
            if (mate_cand_ready) {
              for (uint32_t& x : mate_cand) x = (uint32_t)order[x];
              shn_kp_destroy(kp); kp = nullptr;
              if (dbgk) fprintf(stderr, "[mbgraph]   kp (device, resident) node text %.3f s scan + search %.3f s records %.3f s left to the host + mate pairs %.3f s  (%zu bases, %zu reads, %zu with records, %zu left, %zu node pairs)\n",
                                tk1 - tk0, tk2 - tk1, tk3 - tk2, nowk() - tk3, nb.size(), n_rd(), rec_cnt.size() / 2, left.size(), mate_cand.size() / 2);
              return;
            }
            // (the pairs could not be made on the device: the host's pass needs first / last per read -- start over with host arrays)
            known_paths.clear(); known_edges.clear(); n_known = 0;
          }
          if (kp) { shn_kp_destroy(kp); kp = nullptr; }
          if ((attrs_rc = need_host_attrs())) return;
          if (!ensure_dreads(&d_reads)) { attrs_rc = shn_fail(SHN_ERR_INTERNAL, "find_known_paths: the distinct reads could not be gathered again"); return; }
          tk1 = nowk();
        }
        const bool kp_dev_search = !(getenv("SHN_GRAPH_KP_SEARCH") && getenv("SHN_GRAPH_KP_SEARCH")[0] == '0');
        const int rcs = kp_dev_search
            ? shn_known_paths_search(ctx, d_reads, K, (const uint8_t*)nb.data(), noff.data(), order.size(), eoff.data(), edst.data(), eov.data(), st.data(),
                                     nd.data(), no.data(), precs.data(), precs.size(), &precs_used)
            : shn_known_paths_scan(ctx, d_reads, K, (const uint8_t*)nb.data(), noff.data(), order.size(), st.data(), nd.data(), no.data());
        release_gpu();
        tk2 = nowk();
        if (rcs == 0) {
          // occurrences (node, offset), in index order, of the first K-mers of the reads left to search whose K-mer occurs more
          // than once (state 2: none in most partitions -- the K-mers of a de Bruijn graph are distinct until bridging copies nodes)
          std::unordered_map<uint64_t, std::vector<std::pair<int, int>>> occ_of;
          std::vector<uint64_t> bits(1u << 12, 0);                        // 2^18-bit filter in front of the map
          size_t n_slow = 0, n_multi = 0;
          {
            // the reads inside one node get that node as first / last (a pass over every read: on the threads that are free right now)
            const size_t nr = n_rd();
            const unsigned want0 = (unsigned)std::max<size_t>(1, std::min<size_t>(8, nr >> 19));
            const unsigned nt0 = want0 > 1 ? (unsigned)g_host_threads.take_free((int)want0, g_partitions_running.load() - 1) : 1u;
            struct GiveBack0 { unsigned n; ~GiveBack0() { if (n) g_host_threads.release((int)n); } } give_back0{want0 > 1 ? nt0 : 0u};
            std::vector<size_t> slow_of(nt0, 0);
            std::vector<std::vector<uint32_t>> multi_of(nt0);
            auto settle = [&](unsigned t, size_t lo, size_t hi) {
              size_t ns = 0;
              for (size_t r = lo; r < hi; r++) {
                const uint8_t v = st[r];
                if (v == 1) { const int n = order[nd[r]]; rfirst[r] = n; rlast[r] = n; rhas[r] = 1; }
                else if (v == 3) ns++;
                else if (v == 2) { ns++; multi_of[t].push_back((uint32_t)r); }
              }
              slow_of[t] = ns;
            };
            if (nt0 <= 1) settle(0, 0, nr);
            else {
              std::vector<std::thread> th;
              for (unsigned t = 0; t < nt0; t++) th.emplace_back(settle, t, nr * t / nt0, nr * (t + 1) / nt0);
              for (auto& x : th) x.join();
            }
            for (unsigned t = 0; t < nt0; t++) {
              n_slow += slow_of[t];
              for (uint32_t r : multi_of[t]) {
                uint64_t key;
                n_multi++;
                if (key_at(rstr((int)r), 0, key)) { occ_of[key]; const uint64_t h = fm_mix(key) >> 46; bits[h >> 6] |= 1ULL << (h & 63); }
              }
            }
          }
          if (n_multi)
            for (int n : order) {
              const std::string& b = bases[n];
              uint64_t key = 0;
              for (int i = 0; i < (int)b.size(); i++) {
                key = ((key << 2) | (uint64_t)base_code(b[i])) & mask;
                if (i + 1 < K) continue;
                const uint64_t h = fm_mix(key) >> 46;
                if (!((bits[h >> 6] >> (h & 63)) & 1)) continue;
                auto it = occ_of.find(key);
                if (it != occ_of.end()) it->second.push_back({n, i - K + 1});
              }
            }
          tk3 = nowk();
          // The reads left to search, on host threads (slices of whole 64-read words: the lazily decoded text keeps one done bit per
          // read).  Per thread: the sums per edge and the paths seen go through hash tables while the reads go by (10^5 reads of a highly
          // expressed transcript name the same few edges and paths) and are merged afterwards; the copy counts of reads are whole
          // numbers, so neither the order of the additions nor the cut into slices shows in the sums; a read's first / last node is its own.
          struct Local {
            FlatSum edge_sum;
            std::unordered_map<uint64_t, std::vector<std::vector<int>>> path_seen;
            std::vector<std::vector<int>> fresh;                           // distinct paths of this slice
            int cntp = 0;
          };
          auto search_slice = [&](size_t lo, size_t hi, Local& L) {
            std::vector<std::vector<int>> paths;
            std::vector<int> cur;
            std::vector<std::pair<int, int>> one(1);
            static const std::vector<std::pair<int, int>> none;
            for (size_t r = lo; r < hi; r++) {
              if (st[r] < 2 || st[r] == 4) continue;            // (4: searched on the device, its paths come in the records)
              const RStr rb = rstr((int)r);
              uint64_t key;
              if (!key_at(rb, 0, key)) continue;
              const std::vector<std::pair<int, int>>* occs = &one;
              if (st[r] == 3) one[0] = {order[nd[r]], (int)no[r]};
              else { auto it = occ_of.find(key); occs = it == occ_of.end() ? &none : &it->second; }
              for (const auto& oc : *occs) {
                const int sn = oc.first, so = oc.second;
                if (!compare(rb, 0, bases[sn], so)) continue;
                if (rb.size() <= bases[sn].size() - (size_t)so) { rfirst[r] = sn; rlast[r] = sn; rhas[r] = 1; continue; }
                paths.clear(); cur.clear();
                search_sequence(rb, 0, sn, so, 30, cur, paths);
                for (auto& p : paths) {
                  rfirst[r] = p.front(); rlast[r] = p.back(); rhas[r] = 1;
                  for (size_t j = 0; j + 1 < p.size(); j++) L.edge_sum.add(((uint64_t)(uint32_t)p[j] << 32) | (uint32_t)p[j + 1], rcc[r]);
                  if (p.size() > 2) {
                    L.cntp++;
                    uint64_t h = 0xcbf29ce484222325ULL;
                    for (int v : p) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001b3ULL;
                    auto& lst = L.path_seen[h];
                    bool have = false;
                    for (const auto& q : lst) if (q == p) { have = true; break; }
                    if (!have) { lst.push_back(p); L.fresh.push_back(p); }
                  }
                }
              }
            }
          };
          const size_t words = (n_rd() + 63) / 64;
          const unsigned want = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, std::max(1, shn_host_cpus())), n_slow >> 12));
          const unsigned nts = want > 1 ? (unsigned)g_host_threads.take_free((int)want, g_partitions_running.load() - 1) : 1u;      // (what is free right now: no waiting, no overdraft)
          struct GiveBack { unsigned n; ~GiveBack() { if (n) g_host_threads.release((int)n); } } give_back{want > 1 ? nts : 0u};
          std::vector<Local> locals(nts);
          if (nts <= 1) search_slice(0, n_rd(), locals[0]);
          else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nts; t++)
              th.emplace_back(search_slice, std::min(n_rd(), words * t / nts * 64), std::min(n_rd(), words * (t + 1) / nts * 64), std::ref(locals[t]));
            for (auto& x : th) x.join();
          }
          int cntp = 0;
          for (Local& L : locals) {
            cntp += L.cntp;
            L.edge_sum.each([&](uint64_t k, double v) { known_edges[{(int)(uint32_t)(k >> 32), (int)(uint32_t)k}] += v; });
            for (auto& p : L.fresh) known_paths.insert(std::move(p));
          }
          // the reads the device searched (state 4): their paths as it enumerated them, a read's records in the recursion's order (the
          // last one names its first / last node, mbgraph.py:1379-1384); the sums are of whole numbers, so no order shows in them
          {
            FlatSum edge_sum(4096);
            std::unordered_map<uint64_t, std::vector<std::vector<int>>> dev_seen;
            std::vector<int> pth;
            for (uint64_t at = 0; at + 2 <= precs_used;) {
              const int32_t r = precs[at], len = precs[at + 1];
              if (len <= 0 || at + 2 + (uint64_t)len > precs_used) break;
              pth.resize((size_t)len);
              for (int32_t j = 0; j < len; j++) pth[(size_t)j] = order[(size_t)precs[at + 2 + (uint64_t)j]];
              at += 2 + (uint64_t)len;
              rfirst[r] = pth.front(); rlast[r] = pth.back(); rhas[r] = 1;
              for (size_t j = 0; j + 1 < pth.size(); j++) edge_sum.add(((uint64_t)(uint32_t)pth[j] << 32) | (uint32_t)pth[j + 1], rcc[r]);
              if (pth.size() > 2) {
                cntp++;
                // (the 10^4-10^5 reads of a highly expressed transcript name the same few paths: a hash table in front of the ordered set)
                uint64_t h = 0xcbf29ce484222325ULL;
                for (int v : pth) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001b3ULL;
                auto& lst = dev_seen[h];
                bool have = false;
                for (const auto& q : lst) if (q == pth) { have = true; break; }
                if (!have) { lst.push_back(pth); known_paths.insert(pth); }
              }
            }
            edge_sum.each([&](uint64_t k, double v) { known_edges[{(int)(uint32_t)(k >> 32), (int)(uint32_t)k}] += v; });
          }
          n_known = cntp;
          if (dbgk) fprintf(stderr, "[mbgraph]   kp (device) node text %.3f s scan %.3f s slow index %.3f s search %.3f s  (%zu bases, %zu reads, %zu slow)\n", tk1 - tk0,
                            tk2 - tk1, tk3 - tk2, nowk() - tk3, nb.size(), n_rd(), n_slow);
          return;
        }
      } }
    tk0 = nowk();
    if ((attrs_rc = need_host_attrs())) return;          // (the host form works on per-read arrays of its own)
    ensure_all_text();                                   // (the host form reads every read's text, on several threads)
    std::vector<std::pair<uint64_t, std::pair<int, int>>> items;
    for (int n : order) {
      const std::string& b = bases[n];
      uint64_t key = 0;
      int valid = 0;
      for (int i = 0; i < (int)b.size(); i++) {
        int c = base_code(b[i]);
        if (c < 0) { valid = 0; key = 0; continue; }
        key = ((key << 2) | (uint64_t)c) & mask;
        if (++valid >= K) items.push_back({key, {n, i - K + 1}});
      }
    }
    SeedIndex si;
    si.build(items);
    std::vector<uint32_t> first(n_rd(), 0), last(n_rd(), 0);     // group id + 1, 0 = absent
    shn_table* tab = nullptr;
    bool done = false;
    tk1 = nowk();
    {
      if (gpu_patterns(si, &d_reads, &tab)) {
        done = shn_seed_ends(ctx, d_reads, K, tab, first.data(), last.data()) == 0;
        shn_table_destroy(tab);
      }
    }
    if (!done)
      for (int r = 0; r < (int)n_rd(); r++) {
        const RStr rb = rstr(r);
        uint64_t key;
        if ((int)rb.size() < K) continue;
        if (key_at(rb, 0, key)) first[r] = (uint32_t)(si.find(key) + 1);
        if (key_at(rb, rb.size() - K, key)) last[r] = (uint32_t)(si.find(key) + 1);
      }
    release_gpu();
    tk2 = nowk();
    int cntp = 0;
    std::vector<std::vector<int>> paths;
    std::vector<int> cur;
    // Nearly every read lies inside one node: its only path is that node, it adds no known edge and no known path.  Those reads
    // are settled on host threads; the reads that cross node boundaries (`slow`) go through search_sequence one after the
    // other in read order, as the sums of known_edges require.
    std::vector<char> slow(n_rd(), 0);
    {
      const unsigned ntp = (unsigned)std::min<size_t>(std::min<size_t>(16, std::max(1, shn_host_cpus() / 2)), std::max<size_t>(1, n_rd() >> 18));
      auto classify = [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; r++) {
          if (!first[r] || !last[r]) continue;
          const RStr rb = rstr((int)r);
          const uint32_t gi = first[r] - 1;
          int fn = -1;
          bool any = false, need = false;
          for (uint32_t q = si.goff[gi]; q < si.goff[gi + 1] && !need; q++) {
            const int sn = si.occ[q].first, so = si.occ[q].second;
            if (!compare(rb, 0, bases[sn], so)) continue;
            if (rb.size() <= bases[sn].size() - (size_t)so) { fn = sn; any = true; }
            else need = true;
          }
          if (need) slow[r] = 1;
          else if (any) { rfirst[r] = fn; rlast[r] = fn; rhas[r] = 1; }
        }
      };
      BudgetGuard budget((int)ntp);
      if (ntp <= 1) classify(0, n_rd());
      else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < ntp; t++) th.emplace_back(classify, n_rd() * t / ntp, n_rd() * (t + 1) / ntp);
        for (auto& x : th) x.join();
      }
    }
    tk3 = nowk();
    size_t n_slow = 0;
    for (int r = 0; r < (int)n_rd(); r++) {
      if (!slow[r]) continue;
      n_slow++;
      const RStr rb = rstr(r);
      uint32_t gi = first[r] - 1;
      for (uint32_t q = si.goff[gi]; q < si.goff[gi + 1]; q++) {
        int sn = si.occ[q].first, so = si.occ[q].second;
        if (!compare(rb, 0, bases[sn], so)) continue;
        if (rb.size() <= bases[sn].size() - (size_t)so) {
          // the read lies inside this node (nearly all reads do): its only path is [sn] -- no edges, no known path
          rfirst[r] = sn; rlast[r] = sn; rhas[r] = 1;
          continue;
        }
        paths.clear(); cur.clear();
        search_sequence(rb, 0, sn, so, 30, cur, paths);
        for (auto& p : paths) {
          rfirst[r] = p.front(); rlast[r] = p.back(); rhas[r] = 1;
          for (size_t j = 0; j + 1 < p.size(); j++) known_edges[{p[j], p[j + 1]}] += rcc[r];
          if (p.size() > 2) { known_paths.insert(p); cntp++; }
        }
      }
    }
    n_known = cntp;
    if (dbgk) fprintf(stderr, "[mbgraph]   kp index %.3f s gpu %.3f s classify %.3f s slow %.3f s  (%zu patterns, %zu reads, %zu slow)\n", tk1 - tk0, tk2 - tk1, tk3 - tk2,
                      nowk() - tk3, items.size(), n_rd(), n_slow);
  }
  void find_copy_counts() {
    for (int n : order) {
      double tot = 0;
      bool any = false;
      for (int e : oute[n]) {
        auto it = known_edges.find({es[e], ed[e]});
        double ec = 0;
        if (it != known_edges.end()) { ec = it->second; any = true; }
        tot += ec;
        ecc[e] = ec / (double)std::max(L - ew[e] - 1, 1);
      }
      cc[n] = tot;
      cc_int[n] = any ? 0 : 1;      // Python: `tot` stays the int 0 when no out-edge has a known count
    }
  }
  void mate_search(int n, int goal, int max_len, int min_len, int hops, std::vector<int>& cur, std::vector<std::vector<int>>& out) {
    if (max_len <= 0 || hops <= 0) return;
    if (n == goal && min_len <= 1) { cur.push_back(goal); out.push_back(cur); cur.pop_back(); return; }
    cur.push_back(n);
    for (int e : oute[n]) {
      int nl = (int)bases[n].size() - ew[e];
      mate_search(ed[e], goal, max_len - nl, min_len - nl, hops - 1, cur, out);
    }
    cur.pop_back();
  }
  void find_mate_pairs() {
    std::vector<std::pair<int, int>> pairs;
    std::set<std::pair<int, int>> seen;
    if (dev_attrs && mate_cand_ready) {
      // (made on the device right after the known-paths search, kpaths_gpu.hip: distinct (a, b), not adjacent, in no particular
      // order -- the loop below only counts and fills a set)
      for (size_t i = 0; i + 1 < mate_cand.size(); i += 2) if (seen.insert({(int)mate_cand[i], (int)mate_cand[i + 1]}).second) pairs.push_back({(int)mate_cand[i], (int)mate_cand[i + 1]});
    } else
    for (int r = 0; r < (int)n_rd(); r++) {
      if (rmp[r] != 1 || rmate[r] < 0) continue;
      int m = rmate[r];
      if (!rhas[r] || !rhas[m]) continue;
      int a = rlast[r], b = rfirst[m];
      if (a == b) continue;
      bool adj = false;
      for (int e : oute[a]) if (ed[e] == b) adj = true;
      if (adj) continue;
      if (seen.insert({a, b}).second) pairs.push_back({a, b});
    }
    int nmp = 0;
    for (auto& ab : pairs) {
      int a = ab.first, b = ab.second;
      const int fringe = 1, min_l = 0 - fringe, max_l = 300 - fringe;
      std::vector<std::vector<int>> paths;
      for (int e : oute[a]) {
        std::vector<int> cur(1, a);
        mate_search(ed[e], b, max_l + ew[e], min_l + ew[e], 7, cur, paths);
      }
      if (paths.size() == 1 && paths[0].size() > 2) { nmp++; known_paths.insert(paths[0]); }
    }
    n_mate = nmp;
  }

  int run() {
    const bool dbg = laps;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) { if (dbg) { double t = now(); fprintf(stderr, "[mbgraph] %-22s %8.3f s  nodes=%zu reads=%zu\n", what, t - t0, order.size(), n_rd()); t0 = t; } };
    if (!precondensed) condense_all();
    nodes_after[0] = (int)order.size();
    lap("condense_all");
    destroy_suspicious();
    nodes_after[1] = (int)order.size();
    lap("destroy_suspicious");
    collapse_all();
    nodes_after[2] = (int)order.size();
    lap("collapse_all");
    find_bridging_reads();
    lap("find_bridging_reads");
    int rc = bridge_all();
    if (rc) return rc;
    lap("bridge_all");
    condense_all();
    nodes_after[3] = (int)order.size();
    find_approximate_copy_counts();
    disregard_loops();
    condense_all();
    remove_destroyed();
    lap("condense+copycounts");
    if ((rc = break_cycles())) return rc;
    lap("break_cycles");
    find_approximate_copy_counts();
    find_known_paths();
    if (attrs_rc) return attrs_rc;
    lap("find_known_paths");
    find_copy_counts();
    if (dev_attrs && !mate_cand_ready && (rc = need_host_attrs())) return rc;
    find_mate_pairs();
    lap("copy counts+mates");
    final_nodes = (int)order.size();
    return 0;
  }
};

}  // namespace

extern "C" void shn_graph_destroy(shn_graph* g) { delete g; }
int shn_graph_partitions_running() { return g_partitions_running.load(); }      // (sflow_host.hip: how many cores a sparse-flow call beside the graph stage may take)

// rows: n_rows k1-mers of K+1 bytes each (file order of component{c}k1mers_allowed.dict); reads: ASCII,
// r_off[n_reads+1]; paired: second mate file r2/r2_off with the same count.  Read.L = length of the first read.
// enc: SHN_ENC_ASCII or SHN_ENC_CODES (0..3); rc1/rc2 (optional, one byte per read): 1 = take the reverse
// complement of that read (the strand-doubled view of the packed input, shannon.py:396-424)

static int mbgraph_run_impl(shn_ctx* ctx, int K, const uint8_t* rows, uint64_t n_rows, const shn_unitigs* ug, uint32_t part, const uint8_t* r1,
                           const uint64_t* r1_off, const uint8_t* r2, const uint64_t* r2_off, uint64_t n_reads, int paired, int enc,
                           const uint8_t* rc1, const uint8_t* rc2, shn_graph** out, const shn_reads* src_a = nullptr, const shn_reads* src_b = nullptr,
                           const uint32_t* didx = nullptr, const uint8_t* host_a = nullptr, const uint8_t* host_b = nullptr, const uint32_t* d_didx = nullptr);
extern "C" int shn_reads_dedup(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* didx, uint64_t n, int paired,
                               uint64_t* n_distinct, uint32_t* slot_out, uint32_t* count_out, int32_t* mate_out, uint8_t* role_out);


extern "C" int shn_mbgraph_run(shn_ctx* ctx, int K, const uint8_t* rows, uint64_t n_rows, const uint8_t* r1, const uint64_t* r1_off,
                               const uint8_t* r2, const uint64_t* r2_off, uint64_t n_reads, int paired, int enc, const uint8_t* rc1,
                               const uint8_t* rc2, shn_graph** out) {
  return mbgraph_run_impl(ctx, K, rows, n_rows, nullptr, 0, r1, r1_off, r2, r2_off, n_reads, paired, enc, rc1, rc2, out);
}
// The same with the partition's K-mer graph already contracted on the GPU (shn_unitigs_build, partition `part` of `ug`).
// rows / n_rows (optional): the partition's k1-mers as for shn_mbgraph_run -- needed when the partition holds a cycle of
// condensable edges (left to the sequential code) and for the development check SHN_GRAPH_CHECK=1 (both ways, compared).
extern "C" int shn_mbgraph_run_unitigs(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const uint8_t* r1,
                                       const uint64_t* r1_off, const uint8_t* r2, const uint64_t* r2_off, uint64_t n_reads, int paired, int enc,
                                       const uint8_t* rc1, const uint8_t* rc2, shn_graph** out) {
  if (!ug || part >= ug->n_parts) return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run_unitigs: bad unitigs / partition");
  return mbgraph_run_impl(ctx, ug->K, rows, n_rows, ug, part, r1, r1_off, r2, r2_off, n_reads, paired, enc, rc1, rc2, out);
}

// The same again with the partition's reads known as rows of the resident input: src_a / src_b = the packed read sets of the run
// (src_b NULL for single-end), didx[i] = doubled read index of read i (shn_route_reads' numbering: SE d < N -> R[d], d >= N ->
// RC(R[d-N]); PE d < N -> (R1[d], RC(R1[d])), d >= N -> (RC(R2[d-N]), R2[d-N])).  The read text (r1 / r2) is still what the host
// side of the stage works on; the device copy of the distinct reads is gathered from the resident sets instead of uploaded.
extern "C" int shn_mbgraph_run_resident(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                                        const shn_reads* src_b, const uint32_t* didx, const uint8_t* r1, const uint64_t* r1_off, const uint8_t* r2,
                                        const uint64_t* r2_off, uint64_t n_reads, int paired, int enc, const uint8_t* rc1, const uint8_t* rc2,
                                        shn_graph** out) {
  if (!ug || part >= ug->n_parts) return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run_resident: bad unitigs / partition");
  return mbgraph_run_impl(ctx, ug->K, rows, n_rows, ug, part, r1, r1_off, r2, r2_off, n_reads, paired, enc, rc1, rc2, out, src_a, src_b, didx);
}

// The partition's reads named only by their rows: host_a / host_b = the run's reads as host code matrices (uint8 codes 0-3,
// [n_reads of the set][read length], the same reads as src_a / src_b hold packed on the device), didx as above.  The distinct
// reads are found on the device (shn_reads_dedup) and only their text is decoded on the host, straight from the matrices: no
// per-partition copy of the routed reads exists anywhere.
static int mbgraph_run_rows_impl(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                                 const shn_reads* src_b, const uint8_t* host_a, const uint8_t* host_b, const uint32_t* didx, const uint32_t* d_didx,
                                 uint64_t n_reads, int paired, shn_graph** out) {
  if (!ug || part >= ug->n_parts) return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run_rows: bad unitigs / partition");
  if (!ctx || !src_a || !host_a || (n_reads && !didx && !d_didx) || (paired && (!src_b || !host_b)) || !src_a->fixed_len ||
      (paired && src_b->fixed_len != src_a->fixed_len))
    return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run_rows: needs a context, fixed-length resident read sets and their host matrices");
  const bool dbg = getenv("SHN_GRAPH_LAPS") && n_reads >= strtoull(getenv("SHN_GRAPH_LAPS"), nullptr, 10);
  const double t0 = dbg ? Graph::tnow() : 0.0;
  const int rc = mbgraph_run_impl(ctx, ug->K, rows, n_rows, ug, part, nullptr, nullptr, nullptr, nullptr, n_reads, paired, SHN_ENC_CODES, nullptr, nullptr, out,
                                  src_a, src_b, didx, host_a, host_b, d_didx);
  if (dbg) fprintf(stderr, "[mbgraph] the call (partition %u), its clean-up included %8.3f s\n", part, Graph::tnow() - t0);
  return rc;
}
extern "C" int shn_mbgraph_run_rows(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                                    const shn_reads* src_b, const uint8_t* host_a, const uint8_t* host_b, const uint32_t* didx, uint64_t n_reads,
                                    int paired, shn_graph** out) {
  return mbgraph_run_rows_impl(ctx, ug, part, rows, n_rows, src_a, src_b, host_a, host_b, didx, nullptr, n_reads, paired, out);
}
// The same with the partition's routed reads named a second time by where they already lie on the device: entries [route_lo,
// route_lo + n_reads) of the routes shn_route_reads left there (didx = the same indices on the host) -- the duplicate search then
// reads them in place instead of uploading the list again (0.5 GB per step at BASELINE configs[2], in 111 pieces).
extern "C" int shn_mbgraph_run_routes(shn_ctx* ctx, const shn_unitigs* ug, uint32_t part, const uint8_t* rows, uint64_t n_rows, const shn_reads* src_a,
                                      const shn_reads* src_b, const uint8_t* host_a, const uint8_t* host_b, const uint32_t* didx, const shn_routes* routes,
                                      uint64_t route_lo, uint64_t n_reads, int paired, shn_graph** out) {
  if (!routes) return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run_routes: routes is NULL");
  const uint32_t* d = nullptr;
  int rc = shn_routes_device_slice(routes, route_lo, n_reads, &d);
  if (rc) return rc;
  return mbgraph_run_rows_impl(ctx, ug, part, rows, n_rows, src_a, src_b, host_a, host_b, didx, d, n_reads, paired, out);
}

static int mbgraph_run_impl(shn_ctx* ctx, int K, const uint8_t* rows, uint64_t n_rows, const shn_unitigs* ug, uint32_t part, const uint8_t* r1,
                           const uint64_t* r1_off, const uint8_t* r2, const uint64_t* r2_off, uint64_t n_reads, int paired, int enc,
                           const uint8_t* rc1, const uint8_t* rc2, shn_graph** out, const shn_reads* src_a, const shn_reads* src_b,
                           const uint32_t* didx, const uint8_t* host_a, const uint8_t* host_b, const uint32_t* d_didx) {
  if (!out || (n_rows && !rows) || (!host_a && ((n_reads && (!r1 || !r1_off)) || (paired && n_reads && (!r2 || !r2_off)))))
    return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run: NULL argument");
  struct Running { Running() { g_partitions_running.fetch_add(1); } ~Running() { g_partitions_running.fetch_sub(1); } } running;
  Graph g;
  g.ctx = shn_thread_ctx(ctx);
  g.K = K;
  const uint64_t read_len0 = !n_reads ? 0 : host_a ? src_a->fixed_len : (uint64_t)(r1_off[1] - r1_off[0]);
  g.L = n_reads ? (int)read_len0 : -1;
  g.SIZE_THRESHOLD = g.L;
  // SHN_DEBUG: the laps of every partition; SHN_GRAPH_LAPS=n: of the partitions with at least n routed reads
  const bool dbg = getenv("SHN_DEBUG") != nullptr || (getenv("SHN_GRAPH_LAPS") && n_reads >= strtoull(getenv("SHN_GRAPH_LAPS"), nullptr, 10));
  g.laps = dbg;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tt = now();
  // origin of read slot j = i * nm + mate (mate 0 / 1) in the resident input, for the device gather of the distinct reads
  // (the routed list may be given by its place on the device only: a form that wants it on the host fetches it first)
  std::vector<uint32_t> didx_fetched;
  auto need_didx = [&]() -> int {
    if (didx || !d_didx || !n_reads) return 0;
    didx_fetched.resize(n_reads);
    hipError_t e_ = hipSetDevice(ctx->device);
    if (e_ == hipSuccess) e_ = hipMemcpy(didx_fetched.data(), d_didx, n_reads * 4, hipMemcpyDeviceToHost);
    if (e_ != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_mbgraph_run_routes: ") + hipGetErrorString(e_));
    didx = didx_fetched.data();
    return 0;
  };
  const bool resident = ctx && src_a && (didx || d_didx) && src_a->fixed_len && (!paired || (src_b && src_b->fixed_len == src_a->fixed_len)) &&
                        n_reads && read_len0 == src_a->fixed_len && (host_a || getenv("SHN_GRAPH_RESIDENT_READS") == nullptr);
  if (resident) { g.src_a = src_a; g.src_b = paired ? src_b : nullptr; }
  const uint64_t N_in = src_a ? src_a->n_reads : 0;
  auto origin_of = [&](uint64_t j, uint32_t& row, uint8_t& flag) {
    const int nm_ = paired ? 2 : 1;
    const uint64_t i = j / nm_;
    const int mate = (int)(j % nm_);
    const uint64_t d = didx[i];
    const bool second = d >= N_in;
    row = (uint32_t)(second ? d - N_in : d);
    if (!paired) flag = second ? 2 : 0;
    else if (mate == 0) flag = second ? (1 | 2) : 0;              // R1[d] / RC(R2[d-N])
    else flag = second ? 1 : 2;                                     // RC(R1[d]) / R2[d-N]
  };
  uint64_t n_kmer_nodes = 0;
  if (ug && !ug->cyclic[part]) {
    const uint64_t n0 = ug->node_off[part], n1 = ug->node_off[part + 1], e0 = ug->edge_off[part], e1 = ug->edge_off[part + 1];
    g.load_unitigs(ug->bases.data(), ug->base_off.data() + n0, ug->n_len.data() + n0, ug->n_tail_out.data() + n0, n1 - n0,
                   ug->e_src.data() + e0, ug->e_dst.data() + e0, ug->e_out_rank.data() + e0, ug->e_in_rank.data() + e0, e1 - e0);
    g.precondensed = true;
    n_kmer_nodes = ug->n_kmers[part];
    if (getenv("SHN_GRAPH_CHECK")) {
      if (!rows && n_rows == 0 && n1 != n0) return shn_fail(SHN_ERR_ARG, "SHN_GRAPH_CHECK needs the k1-mer rows");
      Graph h;
      h.K = K;
      h.load_k1mers(rows, n_rows);
      const uint64_t hk = h.order.size();
      h.condense_all();
      if (hk != n_kmer_nodes || h.signature() != g.signature()) {
        if (getenv("SHN_GRAPH_CHECK_DUMP")) {
          FILE* fa = fopen((std::string(getenv("SHN_GRAPH_CHECK_DUMP")) + ".host").c_str(), "w"); if (fa) { fputs(h.signature().c_str(), fa); fclose(fa); }
          FILE* fb = fopen((std::string(getenv("SHN_GRAPH_CHECK_DUMP")) + ".gpu").c_str(), "w"); if (fb) { fputs(g.signature().c_str(), fb); fclose(fb); }
        }
        return shn_fail(SHN_ERR_INTERNAL, "SHN_GRAPH_CHECK: GPU unitigs differ from load_k1mers + condense_all (partition " + std::to_string(part) + ": " +
                        std::to_string(n_kmer_nodes) + " / " + std::to_string(hk) + " K-mers, " + std::to_string(g.order.size()) + " / " +
                        std::to_string(h.order.size()) + " nodes)");
      }
    }
  } else {
    if (ug && !rows && ug->node_off[part + 1] != ug->node_off[part]) return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run_unitigs: cyclic partition needs the k1-mer rows");
    g.load_k1mers(rows, n_rows);
    n_kmer_nodes = g.order.size();
  }
  if (dbg) { fprintf(stderr, "[mbgraph] load_k1mers            %8.3f s  rows=%llu\n", now() - tt, (unsigned long long)n_rows); tt = now(); }
  uint64_t cutoff = n_kmer_nodes * 10;
  // Scratch kept between calls (at the read cap these buffers are 100s of MB, and fresh pages cost more than the work done
  // in them): decode buffers and the read arena.  A free list, not thread_local: Python's partition workers are short-lived.
  // (also the per-read arrays of the graph: a partition at the read cap touches ~1.5 GB here, and with 64 partitions starting at
  // once the page faults of fresh memory were most of the largest partition's "load reads".)  The list keeps one object per
  // partition thread; a call takes the smallest one that is large enough, else the largest.
  struct Scratch {
    std::vector<uint64_t> doff, hashes; std::vector<char> text; std::string arena; std::vector<uint32_t> first, cnt, last; std::vector<int32_t> idmap;
    std::vector<double> rcc; std::vector<int> rmate, rmp, rfirst, rlast; std::vector<char> rhas; std::vector<uint64_t> r_hashes, r_off;
    std::vector<uint8_t> role;
    std::vector<uint32_t> origin_row; std::vector<uint8_t> origin_flag;      // (the graph's, kept from call to call like the arrays above)
    // the buffer of the lazily decoded read text: NOT a vector -- it is written piecemeal (a few per cent of the reads are ever
    // decoded), and a vector's resize zero-filled all of it: 0.1 s for the 354 MB of the largest partition whenever it got a
    // scratch object that had served a smaller one (the spread of the graph stage from step to step)
    char* lz_raw = nullptr; size_t lz_cap = 0;
    char* lazy_text(size_t bytes) {
      if (bytes > lz_cap) { free(lz_raw); lz_raw = (char*)malloc(bytes + bytes / 8); lz_cap = lz_raw ? bytes + bytes / 8 : 0; }
      return lz_raw;
    }
    ~Scratch() { free(lz_raw); }
    size_t room() const { return std::max(std::max(text.capacity(), arena.capacity()), lz_cap); }
  };
  static std::mutex scratch_mu;
  static std::vector<Scratch*> scratch_free;
  Scratch* sc = nullptr;
  // the distinct reads found on the device: with the host matrices always, with gathered rows for large sets
  uint64_t bulk_min = 1u << 17;                         // reads from which the duplicates are found in parallel (tests lower it)
  if (getenv("SHN_GRAPH_BULK_MIN")) bulk_min = strtoull(getenv("SHN_GRAPH_BULK_MIN"), nullptr, 10);
  const char* dd_env = getenv("SHN_GRAPH_DEVICE_DEDUP");
  const bool dev_dedup = resident && (host_a || ((!dd_env || dd_env[0] != '0') && enc == SHN_ENC_CODES &&
                                                 std::min<uint64_t>(n_reads, cutoff + 1) * (paired ? 2 : 1) >= bulk_min));
  const size_t need_text = (size_t)std::min<uint64_t>(n_reads, cutoff + 1) * (paired ? 2 : 1) * (size_t)read_len0 / (dev_dedup ? 2 : 1);
  { std::lock_guard<std::mutex> lk(scratch_mu);
    int pick = -1;
    for (size_t i = 0; i < scratch_free.size(); i++) {
      const size_t c = scratch_free[i]->room();
      if (pick < 0) { pick = (int)i; continue; }
      const size_t pc = scratch_free[pick]->room();
      if (pc >= need_text ? (c >= need_text && c < pc) : c > pc) pick = (int)i;
    }
    if (pick >= 0) { sc = scratch_free[pick]; scratch_free.erase(scratch_free.begin() + pick); } }
  if (!sc) sc = new Scratch();
  struct Giveback {
    Scratch* s; Graph* g;
    ~Giveback() {
      g->rindex.arena.clear(); s->arena.swap(g->rindex.arena);
      s->rcc.swap(g->rcc); s->rmate.swap(g->rmate); s->rmp.swap(g->rmp); s->rfirst.swap(g->rfirst); s->rlast.swap(g->rlast); s->rhas.swap(g->rhas);
      s->r_hashes.swap(g->rindex.hashes); s->r_off.swap(g->rindex.off);
      s->origin_row.swap(g->origin_row); s->origin_flag.swap(g->origin_flag);
      std::lock_guard<std::mutex> lk(scratch_mu);
      if (scratch_free.size() < 192) scratch_free.push_back(s); else delete s;
    }
  } giveback{sc, &g};
  g.rindex.arena.swap(sc->arena);
  g.rindex.arena.clear();
  sc->rcc.clear(); sc->rmate.clear(); sc->rmp.clear(); sc->rfirst.clear(); sc->rlast.clear(); sc->rhas.clear(); sc->r_hashes.clear(); sc->r_off.assign(1, 0);
  g.rcc.swap(sc->rcc); g.rmate.swap(sc->rmate); g.rmp.swap(sc->rmp); g.rfirst.swap(sc->rfirst); g.rlast.swap(sc->rlast); g.rhas.swap(sc->rhas);
  g.rindex.hashes.swap(sc->r_hashes); g.rindex.off.swap(sc->r_off);
  g.origin_row.swap(sc->origin_row); g.origin_flag.swap(sc->origin_flag);
  g.origin_row.resize(0); g.origin_flag.resize(0);                      // (capacity kept; empty until a path fills them)
  // the fast form of the rows mode (graph_dev.h): the duplicate search leaves its arrays on the device, the host gets rows + strands
  // (for the lazily decoded text) and nothing else per read.  SHN_GRAPH_DEV_ATTRS=0: the host-array form below.
  const char* dav = getenv("SHN_GRAPH_DEV_ATTRS");
  const bool lazy_ok = host_a && !(getenv("SHN_GRAPH_LAZY_TEXT") && getenv("SHN_GRAPH_LAZY_TEXT")[0] == '0') && src_a && src_a->n_invalid == 0 &&
                       (!paired || (src_b && src_b->n_invalid == 0));
  if (dev_dedup && lazy_ok && g.ctx && K <= 31 && !(dav && dav[0] == '0')) {
    const uint64_t used = std::min<uint64_t>(n_reads, cutoff + 1);
    const uint64_t Lr = read_len0;
    double t_dec = now();
    int rcd = shn_reads_dedup_dev(g.ctx, src_a, paired ? src_b : nullptr, didx, d_didx, used, paired, &g.dd);
    if (rcd) return rcd;
    const uint64_t nd = g.dd->n_distinct;
    g.n_rd_dev = nd; g.dev_attrs = true;
    if (dbg) fprintf(stderr, "[mbgraph]   distinct reads (GPU)   %8.3f s  used=%llu distinct=%llu (attributes stay on the device)\n", now() - t_dec, (unsigned long long)used, (unsigned long long)nd);
    g.lz_a = host_a; g.lz_b = host_b; g.lz_L = (uint32_t)Lr; g.lz_buf = sc->lazy_text(nd * Lr + 1);
    if (!g.lz_buf) return shn_fail(SHN_ERR_NOMEM, "shn_mbgraph_run: out of host memory for the reads' text");
    g.lz_done.assign((nd + 63) / 64, 0);
    g.origin_row.resize(nd); g.origin_flag.resize(nd);
    if ((rcd = shn_dedup_origin(g.dd, g.origin_row.data(), g.origin_flag.data()))) return rcd;
    g.acgt_known = 1;
    if (dbg) fprintf(stderr, "[mbgraph]   + rows of the distinct  %8.3f s\n", now() - t_dec);
  } else if (dev_dedup) {
    { const int rcf = need_didx(); if (rcf) return rcf; }
    const uint64_t used = std::min<uint64_t>(n_reads, cutoff + 1);
    const int nm = paired ? 2 : 1;
    const uint64_t nh = used * nm, Lr = read_len0;
    double t_dec = now();
    std::vector<uint32_t>&slot = sc->first, &cnt = sc->cnt;
    std::vector<int32_t>& mate = sc->idmap;
    std::vector<uint8_t>& role = sc->role;
    slot.resize(nh); cnt.resize(nh); mate.resize(nh); role.resize(nh);
    uint64_t nd = 0;
    int rcd = shn_reads_dedup(g.ctx, src_a, paired ? src_b : nullptr, didx, used, paired, &nd, slot.data(), cnt.data(), mate.data(), role.data());
    if (rcd) return rcd;
    if (dbg) fprintf(stderr, "[mbgraph]   distinct reads (GPU)   %8.3f s  used=%llu distinct=%llu\n", now() - t_dec, (unsigned long long)used, (unsigned long long)nd);
    StringInterner& R = g.rindex;
    R.hashes.assign(nd, 0);
    const char* lzv = getenv("SHN_GRAPH_LAZY_TEXT");
    const bool lazy = host_a && !(lzv && lzv[0] == '0') && src_a->n_invalid == 0 && (!paired || src_b->n_invalid == 0);
    if (lazy) {
      g.lz_a = host_a; g.lz_b = host_b; g.lz_L = (uint32_t)Lr; g.lz_buf = sc->lazy_text(nd * Lr + 1);
      if (!g.lz_buf) return shn_fail(SHN_ERR_NOMEM, "shn_mbgraph_run: out of host memory for the reads' text");
      g.lz_done.assign((nd + 63) / 64, 0);
    }
    R.off.resize(nd + 1);
    if (!lazy && R.arena.capacity() < nd * Lr) { R.arena.reserve(nd * Lr); if (nd * Lr >= (8u << 20)) { const uintptr_t a = ((uintptr_t)R.arena.data() + 4095) & ~(uintptr_t)4095; madvise((void*)a, (nd * Lr) & ~(size_t)4095, MADV_HUGEPAGE); } }
    if (!lazy) R.arena.resize(nd * Lr);
    g.rcc.resize(nd); g.rmate.resize(nd); g.rmp.resize(nd); g.rfirst.assign(nd, -1); g.rlast.assign(nd, -1); g.rhas.assign(nd, 0);
    g.origin_row.resize(nd); g.origin_flag.resize(nd);
    const unsigned hwc = (unsigned)shn_host_cpus();
    unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(32, std::max(1u, hwc / 2)), nd >> 15));
    BudgetGuard budget((int)nt);
    std::atomic<int> non_acgt{0};
    char* arena = lazy ? nullptr : &R.arena[0];
    auto work = [&](uint64_t lo, uint64_t hi) {
      bool bad = false;
      for (uint64_t id = lo; id < hi; id++) {
        const uint64_t j = slot[id];
        uint32_t row; uint8_t fl;
        origin_of(j, row, fl);
        g.origin_row[id] = row; g.origin_flag[id] = fl;
        if (lazy) { g.rcc[id] = (double)cnt[id]; g.rmate[id] = mate[id]; g.rmp[id] = role[id]; continue; }
        const uint8_t* p; bool rc; int e = enc;
        if (host_a) { p = ((fl & 1) ? host_b : host_a) + (uint64_t)row * Lr; rc = (fl & 2) != 0; e = SHN_ENC_CODES; }
        else { const uint64_t i = j / nm; if (j % nm == 0) { p = r1 + r1_off[i]; rc = rc1 && rc1[i]; } else { p = r2 + r2_off[i]; rc = rc2 && rc2[i]; } }
        char* d = arena + id * Lr;
        decode_read(d, p, Lr, e, rc);
        for (uint64_t q = 0; q < Lr; q++) bad |= !(d[q] == 'A' || d[q] == 'C' || d[q] == 'G' || d[q] == 'T');
        R.off[id + 1] = (id + 1) * Lr;
        g.rcc[id] = (double)cnt[id];
        g.rmate[id] = mate[id];
        g.rmp[id] = role[id];
      }
      if (bad) non_acgt.store(1);
    };
    R.off[0] = 0;
    if (nt <= 1) work(0, nd);
    else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; t++) th.emplace_back(work, nd * t / nt, nd * (t + 1) / nt);
      for (auto& t : th) t.join();
    }
    if (non_acgt.load()) return shn_fail(SHN_ERR_ARG, "shn_mbgraph_run: a routed read holds a base outside ACGT (the packed rows cannot tell such reads apart)");
    g.acgt_known = 1;
    R.bulk_loaded = true;
    if (dbg) fprintf(stderr, "[mbgraph]   + text of the distinct %8.3f s  nt=%u (waited %.3f s for threads)\n", now() - t_dec, nt, budget.waited);
  } else {
    uint64_t used = std::min<uint64_t>(n_reads, cutoff + 1);
    size_t bytes = used ? (size_t)(r1_off[used] - r1_off[0]) + (paired ? (size_t)(r2_off[used] - r2_off[0]) : 0) : 0;
    g.rindex.arena.reserve(bytes);
    size_t nr = (size_t)used * (paired ? 2 : 1);
    if (nr < (1u << 17)) g.rindex.reserve(nr);          // (large sets are numbered in bulk and never use the interner's probe table)
    g.rcc.reserve(nr); g.rmate.reserve(nr); g.rmp.reserve(nr); g.rfirst.reserve(nr); g.rlast.reserve(nr); g.rhas.reserve(nr);
  }
  if (!dev_dedup) {
    { const int rcf = need_didx(); if (rcf) return rcf; }
    // decode + hash on several host threads (independent per read), then intern sequentially in file order
    const uint64_t used = std::min<uint64_t>(n_reads, cutoff + 1);
    const int nm = paired ? 2 : 1;
    // (scratch kept per host thread between calls: at the read cap these are 100s of MB, and fresh pages cost more than the decode)
    std::vector<uint64_t>&doff = sc->doff, &hashes = sc->hashes;
    std::vector<char>& text = sc->text;
    doff.resize((size_t)used * nm + 1);
    doff[0] = 0;
    for (uint64_t i = 0; i < used; i++) {
      doff[i * nm + 1] = doff[i * nm] + (r1_off[i + 1] - r1_off[i]);
      if (paired) doff[i * nm + 2] = doff[i * nm + 1] + (r2_off[i + 1] - r2_off[i]);
    }
    auto huge = [](const void* p, size_t bytes) {                 // transparent huge pages for the large scratch buffers (THP in madvise mode)
      if (bytes < (8u << 20)) return;
      const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
      if (e > a) madvise((void*)a, e - a, MADV_HUGEPAGE);
    };
    if (text.size() < doff.back() + 1) { text.clear(); text.reserve(doff.back() + 1); huge(text.data(), text.capacity()); text.resize(doff.back() + 1); }
    if (hashes.capacity() < (size_t)used * nm) { hashes.clear(); hashes.reserve((size_t)used * nm); huge(hashes.data(), hashes.capacity() * 8); }
    hashes.resize((size_t)used * nm);
    if (g.rindex.arena.capacity() >= (8u << 20)) huge(g.rindex.arena.data(), g.rindex.arena.capacity());
    double t_dec = now();
    std::atomic<int> non_acgt{0};
    auto work = [&](uint64_t lo, uint64_t hi) {
      bool bad = false;
      auto check = [&](const char* d, uint64_t n) { for (uint64_t j = 0; j < n; j++) bad |= !(d[j] == 'A' || d[j] == 'C' || d[j] == 'G' || d[j] == 'T'); };
      for (uint64_t i = lo; i < hi; i++) {
        char* d1 = text.data() + doff[i * nm];
        const uint64_t n1 = r1_off[i + 1] - r1_off[i];
        decode_read(d1, r1 + r1_off[i], n1, enc, rc1 && rc1[i]);
        check(d1, n1);
        hashes[i * nm] = StringInterner::hash(d1, n1);
        if (paired) {
          char* d2 = text.data() + doff[i * nm + 1];
          const uint64_t n2 = r2_off[i + 1] - r2_off[i];
          decode_read(d2, r2 + r2_off[i], n2, enc, rc2 && rc2[i]);
          check(d2, n2);
          hashes[i * nm + 1] = StringInterner::hash(d2, n2);
        }
      }
      if (bad) non_acgt.store(1);
    };
    // (8 ranks share a node's cores; partitions running at the same time in this process share this rank's part)
    static std::atomic<int> active_calls{0};
    struct Active { std::atomic<int>& a; int n; Active(std::atomic<int>& x) : a(x), n(++x) {} ~Active() { --a; } } active(active_calls);
    // The partitions of a run differ in size by an order of magnitude and the largest ones are started first (pipeline.py): the
    // stage ends when the largest partition does, so its read set gets threads in proportion to its size (one per 256 Ki
    // reads, at most 32 and at most a quarter of the cores), whatever else is running; small sets share what is left.
    const unsigned hwc = (unsigned)shn_host_cpus();
    unsigned nt = std::min<unsigned>(32, std::max<unsigned>(1, hwc / 8 / (unsigned)std::max(1, active.n / 2)));
    nt = std::max<unsigned>(nt, (unsigned)std::min<uint64_t>(std::min<uint64_t>(32, std::max(1u, hwc / 2)), (used * (paired ? 2 : 1)) >> 18));
    if (getenv("SHN_GRAPH_BULK_MIN")) nt = std::max(nt, 4u);
    if (used * nm < bulk_min) nt = used < 4096 ? 1 : std::min<unsigned>(nt, (unsigned)(used / 2048));   // small sets: a few threads for the decode only
    BudgetGuard budget((int)nt);                          // held until the reads are numbered
    if (nt <= 1) work(0, used);
    else {
      std::vector<std::thread> th;
      const uint64_t per = (used + nt - 1) / nt;
      for (unsigned t = 0; t < nt; t++) { uint64_t lo = t * per, hi = std::min<uint64_t>(used, lo + per); if (lo < hi) th.emplace_back(work, lo, hi); }
      for (auto& t : th) t.join();
    }
    g.acgt_known = non_acgt.load() ? 0 : 1;
    if (dbg) fprintf(stderr, "[mbgraph]   offsets+decode+hash   %8.3f s  used=%llu nt=%u (waited %.3f s for threads)\n", now() - t_dec, (unsigned long long)used, nt, budget.waited);
    const uint64_t nh = used * nm;
    if (nt > 1 && nh >= bulk_min && g.rindex.size() == 0) {
      // Large read sets, all on `nt` host threads: (1) the duplicates -- every thread owns the strings whose hash falls into
      // its shard (private open-addressing table: string -> index of its first occurrence, with its number of occurrences
      // and its last occurrence); (2) ids in file order of first occurrence = a prefix sum over the "first occurrence"
      // flags, the strings copied to their place in the arena in parallel; (3) every read's id; (4) mates: interning one
      // pair after the other leaves every read with the role and mate of its LAST occurrence.  Same ids, counts and mates
      // as reading one read at a time (test_native_graph_stage_parallel_read_dedup).
      std::vector<uint32_t>&first = sc->first, &cnt = sc->cnt, &last = sc->last;
      first.resize(nh); cnt.resize(nh); last.resize(nh);
      auto run_threads = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(fn, t);
        for (auto& t : th) t.join();
      };
      run_threads([&](unsigned t) {
        size_t cap = 1024;                                      // shards of a hash are even: 2.5x the mean share is ample
        while (cap < (nh / nt + 1) * 5 / 2) cap <<= 1;
        std::vector<uint32_t> tab(cap, 0xFFFFFFFFu);
        uint64_t used_slots = 0;
        for (uint64_t j = 0; j < nh; j++) {
          const uint64_t h = hashes[j];
          if (((h >> 40) % nt) != t) continue;
          if (used_slots * 10 > cap * 8) {                       // (a pathological hash distribution: grow and re-insert)
            std::vector<uint32_t> old;
            old.swap(tab);
            cap <<= 1;
            tab.assign(cap, 0xFFFFFFFFu);
            for (uint32_t q : old) if (q != 0xFFFFFFFFu) { size_t sl = (size_t)hashes[q] & (cap - 1); while (tab[sl] != 0xFFFFFFFFu) sl = (sl + 1) & (cap - 1); tab[sl] = q; }
          }
          const size_t m = cap - 1;
          const char* p = text.data() + doff[j];
          const uint64_t n = doff[j + 1] - doff[j];
          size_t sl = (size_t)h & m;
          while (true) {
            const uint32_t q = tab[sl];
            if (q == 0xFFFFFFFFu) { tab[sl] = (uint32_t)j; first[j] = (uint32_t)j; cnt[j] = 1; last[j] = (uint32_t)j; used_slots++; break; }
            if (hashes[q] == h && doff[q + 1] - doff[q] == n && memcmp(text.data() + doff[q], p, n) == 0) { first[j] = q; cnt[q]++; last[q] = (uint32_t)j; break; }
            sl = (sl + 1) & m;
          }
        }
      });
      if (dbg) fprintf(stderr, "[mbgraph]   + duplicates found     %8.3f s  used=%llu\n", now() - t_dec, (unsigned long long)used);
      std::vector<int32_t>& idmap = sc->idmap;
      idmap.resize(nh);
      StringInterner& R = g.rindex;
      // (2) chunk c of the reads: how many first occurrences, how many bytes
      std::vector<uint64_t> nf(nt + 1, 0), nbytes(nt + 1, 0);
      auto lo_of = [&](unsigned c) { return nh * c / nt; };
      run_threads([&](unsigned c) {
        uint64_t f = 0, by = 0;
        for (uint64_t j = lo_of(c); j < lo_of(c + 1); j++) if (first[j] == (uint32_t)j) { f++; by += doff[j + 1] - doff[j]; }
        nf[c + 1] = f; nbytes[c + 1] = by;
      });
      for (unsigned c = 0; c < nt; c++) { nf[c + 1] += nf[c]; nbytes[c + 1] += nbytes[c]; }
      const uint64_t nd = nf[nt];
      R.hashes.resize(nd);
      R.off.resize(nd + 1);
      R.off[0] = 0;
      R.arena.resize(nbytes[nt]);
      g.rcc.resize(nd); g.rmate.resize(nd, -1); g.rmp.resize(nd, 0); g.rfirst.resize(nd, -1); g.rlast.resize(nd, -1); g.rhas.resize(nd, 0);
      if (resident) { g.origin_row.resize(nd); g.origin_flag.resize(nd); }
      char* arena = &R.arena[0];
      run_threads([&](unsigned c) {
        uint64_t id = nf[c], at = nbytes[c];
        for (uint64_t j = lo_of(c); j < lo_of(c + 1); j++) {
          if (first[j] != (uint32_t)j) continue;
          const uint64_t n = doff[j + 1] - doff[j];
          memcpy(arena + at, text.data() + doff[j], n);
          at += n;
          R.hashes[id] = hashes[j];
          R.off[id + 1] = at;
          g.rcc[id] = (double)cnt[j];
          idmap[j] = (int32_t)id;
          if (resident) origin_of(j, g.origin_row[id], g.origin_flag[id]);
          id++;
        }
      });
      run_threads([&](unsigned c) {                                  // (3) (a first occurrence precedes its duplicates, all are numbered by now)
        for (uint64_t j = lo_of(c); j < lo_of(c + 1); j++) if (first[j] != (uint32_t)j) idmap[j] = idmap[first[j]];
      });
      if (paired)
        run_threads([&](unsigned c) {                                // (4)
          for (uint64_t j = lo_of(c); j < lo_of(c + 1); j++) {
            if (first[j] != (uint32_t)j) continue;
            const uint32_t l = last[j];
            const int32_t id = idmap[j];
            g.rmp[id] = (l & 1) ? 2 : 1;
            g.rmate[id] = idmap[l ^ 1u];
          }
        });
      R.bulk_loaded = true;                 // (its probe table was bypassed: no interning by string after this)
      if (dbg) fprintf(stderr, "[mbgraph]   + numbered in order    %8.3f s  used=%llu\n", now() - t_dec, (unsigned long long)used);
    } else
    for (uint64_t i = 0; i < used; i++) {
      auto note = [&](int r, uint64_t j) {
        if (resident && (size_t)r == g.origin_row.size()) { uint32_t row; uint8_t fl; origin_of(j, row, fl); g.origin_row.push_back(row); g.origin_flag.push_back(fl); }
      };
      int a = g.add_read(text.data() + doff[i * nm], doff[i * nm + 1] - doff[i * nm], hashes[i * nm]);
      note(a, i * nm);
      if (paired) {
        int b = g.add_read(text.data() + doff[i * nm + 1], doff[i * nm + 2] - doff[i * nm + 1], hashes[i * nm + 1]);
        note(b, i * nm + 1);
        g.rmp[a] = 1; g.rmp[b] = 2; g.rmate[a] = b; g.rmate[b] = a;
      }
    }
  }
  if (dbg) { fprintf(stderr, "[mbgraph] load reads             %8.3f s  reads=%llu distinct=%zu\n", now() - tt, (unsigned long long)n_reads, g.n_rd()); tt = now(); }
  int rc = g.run();
  g.release_gpu();
  tt = now();
  if (rc) return rc;
  // ---- output_components (add_component mbgraph.py:691-709, topological_sort :711-732, P2)
  shn_graph* o = new shn_graph();
  o->s_off.push_back(0); o->n_off.push_back(0); o->p_off.push_back(0);
  o->comp_node_off.push_back(0); o->comp_edge_off.push_back(0); o->comp_path_off.push_back(0);
  std::map<int, std::vector<const std::vector<int>*>> by_start;
  for (auto& p : g.known_paths) by_start[p[0]].push_back(&p);
  for (int src : g.order) {
    if (g.dead[src]) continue;
    std::set<int> seen;
    std::set<int> edges;
    std::vector<int> queue(1, src);
    while (!queue.empty()) {
      int n = queue.back(); queue.pop_back();
      if (seen.count(n)) continue;
      seen.insert(n);
      for (int e : g.oute[n]) edges.insert(e);
      for (int e : g.oute[n]) queue.push_back(g.ed[e]);
      for (int e : g.ine[n]) queue.push_back(g.es[e]);
    }
    std::set<int> added;
    std::vector<int> topo, fringe;
    for (int n : seen) if (g.ine[n].empty()) fringe.push_back(n);      // `seen` iterates in creation (id) order
    while (!fringe.empty()) {
      int v = fringe.back(); fringe.pop_back();
      if (added.count(v)) continue;
      added.insert(v);
      topo.push_back(v);
      for (int e : g.oute[v]) {
        int n = g.ed[e];
        bool all = true;
        for (int pe : g.ine[n]) if (!added.count(g.es[pe])) all = false;
        if (all) fringe.push_back(n);
      }
    }
    if (topo.size() == 1) {
      g.hash[src] = -1;
      o->s_bases += g.bases[src];
      o->s_off.push_back(o->s_bases.size());
      o->s_cc.push_back(g.cc[src]); o->s_norm.push_back(g.norm[src]);
      g.dead[src] = 1;
      continue;
    }
    for (size_t h = 0; h < topo.size(); h++) { g.hash[topo[h]] = (int)h; g.dead[topo[h]] = 1; }
    for (int n : topo) {
      o->n_bases += g.bases[n];
      o->n_off.push_back(o->n_bases.size());
      o->n_cc.push_back(g.cc[n]); o->n_norm.push_back(g.norm[n]); o->n_cc_int.push_back((uint8_t)g.cc_int[n]);
    }
    o->comp_node_off.push_back(o->n_cc.size());
    for (int n : topo) {
      auto it = by_start.find(n);
      if (it == by_start.end()) continue;
      std::vector<std::vector<int>> ps;
      for (auto* p : it->second) { std::vector<int> h; for (int x : *p) h.push_back(g.hash[x]); ps.push_back(h); }
      std::sort(ps.begin(), ps.end());
      for (auto& h : ps) { for (int x : h) o->p_ids.push_back(x); o->p_off.push_back(o->p_ids.size()); }
    }
    o->comp_path_off.push_back(o->p_off.size() - 1);
    std::vector<std::array<double, 5>> el;
    for (int e : edges) if (g.ecc[e] > 0)
      el.push_back({(double)g.hash[g.es[e]], (double)g.hash[g.ed[e]], (double)g.ew[e], g.ecc[e], (double)std::max(g.L - g.ew[e] - 1, 0)});
    std::stable_sort(el.begin(), el.end(), [](const std::array<double, 5>& a, const std::array<double, 5>& b) {
      if (a[0] != b[0]) return a[0] < b[0];
      if (a[1] != b[1]) return a[1] < b[1];
      return a[2] < b[2];
    });
    for (auto& x : el) { o->e_in.push_back((int)x[0]); o->e_out.push_back((int)x[1]); o->e_w.push_back((int)x[2]); o->e_cc.push_back(x[3]); o->e_norm.push_back(x[4]); }
    o->comp_edge_off.push_back(o->e_in.size());
  }
  if (dbg) fprintf(stderr, "[mbgraph] output_components      %8.3f s\n", now() - tt);
  for (int i = 0; i < 4; i++) o->info.push_back(g.nodes_after[i]);
  o->info.push_back(g.final_nodes); o->info.push_back(g.n_known); o->info.push_back(g.n_mate);
  o->info.push_back((int)g.bridged_log.size());
  for (int b : g.bridged_log) o->info.push_back(b);
  *out = o;
  return SHN_OK;
}

// sizes: [n_singles, s_bases, n_comps, n_nodes, n_bases, n_edges, n_paths, n_path_ids, n_info]
extern "C" int shn_graph_sizes(const shn_graph* g, uint64_t* sizes) {
  if (!g || !sizes) return shn_fail(SHN_ERR_ARG, "shn_graph_sizes: NULL argument");
  sizes[0] = g->s_cc.size(); sizes[1] = g->s_bases.size(); sizes[2] = g->comp_node_off.size() - 1; sizes[3] = g->n_cc.size();
  sizes[4] = g->n_bases.size(); sizes[5] = g->e_in.size(); sizes[6] = g->p_off.size() - 1; sizes[7] = g->p_ids.size();
  sizes[8] = g->info.size();
  return SHN_OK;
}

extern "C" int shn_graph_export(const shn_graph* g, uint64_t* s_off, uint8_t* s_bases, double* s_cc, double* s_norm,
                                uint64_t* comp_node_off, uint64_t* comp_edge_off, uint64_t* comp_path_off, uint64_t* n_off,
                                uint8_t* n_bases, double* n_cc, uint8_t* n_cc_int, double* n_norm, int32_t* e_in, int32_t* e_out,
                                int32_t* e_w, double* e_cc, double* e_norm, uint64_t* p_off, int32_t* p_ids, int32_t* info) {
  if (!g) return shn_fail(SHN_ERR_ARG, "shn_graph_export: NULL graph");
#define CP(dst, v) if (dst && !(v).empty()) memcpy(dst, (v).data(), (v).size() * sizeof((v)[0]))
  CP(s_off, g->s_off); CP(s_bases, g->s_bases); CP(s_cc, g->s_cc); CP(s_norm, g->s_norm);
  CP(comp_node_off, g->comp_node_off); CP(comp_edge_off, g->comp_edge_off); CP(comp_path_off, g->comp_path_off);
  CP(n_off, g->n_off); CP(n_bases, g->n_bases); CP(n_cc, g->n_cc); CP(n_cc_int, g->n_cc_int); CP(n_norm, g->n_norm);
  CP(e_in, g->e_in); CP(e_out, g->e_out); CP(e_w, g->e_w); CP(e_cc, g->e_cc); CP(e_norm, g->e_norm);
  CP(p_off, g->p_off); CP(p_ids, g->p_ids); CP(info, g->info);
#undef CP
  return SHN_OK;
}

// The inverse of shn_graph_export: a shn_graph from flattened component tables (sizes[9] as shn_graph_sizes; arrays as
// shn_graph_export, info may be NULL) -- lets the native sparse-flow stage run on tables that did not come from
// shn_mbgraph_run (the reference's own nodes / edges / paths files in the tests).
extern "C" int shn_graph_from_tables(const uint64_t* sizes, const uint64_t* s_off, const uint8_t* s_bases, const double* s_cc, const double* s_norm,
                                     const uint64_t* comp_node_off, const uint64_t* comp_edge_off, const uint64_t* comp_path_off, const uint64_t* n_off,
                                     const uint8_t* n_bases, const double* n_cc, const uint8_t* n_cc_int, const double* n_norm, const int32_t* e_in,
                                     const int32_t* e_out, const int32_t* e_w, const double* e_cc, const double* e_norm, const uint64_t* p_off,
                                     const int32_t* p_ids, shn_graph** out) {
  if (!sizes || !out) return shn_fail(SHN_ERR_ARG, "shn_graph_from_tables: NULL argument");
  shn_graph* g = new shn_graph();
  const uint64_t ns = sizes[0], sb = sizes[1], nc = sizes[2], nn = sizes[3], nb = sizes[4], ne = sizes[5], np = sizes[6], npid = sizes[7];
  g->s_off.assign(s_off, s_off + ns + 1); g->s_bases.assign((const char*)s_bases, sb); g->s_cc.assign(s_cc, s_cc + ns); g->s_norm.assign(s_norm, s_norm + ns);
  g->comp_node_off.assign(comp_node_off, comp_node_off + nc + 1); g->comp_edge_off.assign(comp_edge_off, comp_edge_off + nc + 1);
  g->comp_path_off.assign(comp_path_off, comp_path_off + nc + 1);
  g->n_off.assign(n_off, n_off + nn + 1); g->n_bases.assign((const char*)n_bases, nb); g->n_cc.assign(n_cc, n_cc + nn); g->n_norm.assign(n_norm, n_norm + nn);
  g->n_cc_int.assign(n_cc_int, n_cc_int + nn);
  g->e_in.assign(e_in, e_in + ne); g->e_out.assign(e_out, e_out + ne); g->e_w.assign(e_w, e_w + ne); g->e_cc.assign(e_cc, e_cc + ne); g->e_norm.assign(e_norm, e_norm + ne);
  g->p_off.assign(p_off, p_off + np + 1); g->p_ids.assign(p_ids, p_ids + npid);
  *out = g;
  return SHN_OK;
}
