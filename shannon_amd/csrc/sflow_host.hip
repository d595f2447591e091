// Sparse-flow transcript reconstruction of all components of all partitions, native host code over the batched LP
// kernel (rows a25-a30): the same algorithm as shannon_amd/sparse_flow.py (which stays as the readable mirror of the
// reference's algorithm_SF.py:74-613 / path_decompose_sparse.py:15-193 and is cross-checked against this in the tests).
// At 20,000 genes the Python form spent ~4 s per step on 40,000 components plus the conversion of every graph into Python
// lists; here the components are state machines over the flattened graphs (shn_graph) that stop at each node decomposition
// needing LP trials, all pending decompositions go to the device in one batch (shn_lp_solve_batch), and the FASTA text
// of every partition comes out as one string.
#include "common.h"
#include "graph_result.h"
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <vector>
#include <memory>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {

const int PATH_SPARSITY = 10;        // algorithm_SF.py:31

// Python's repr(float) (the reference prints str(float); shannon_amd/sparse_flow.py prints the same): shortest digits that
// round-trip, fixed notation for 1e-4 <= |x| < 1e16, else d[.ddd]e+XX with at least two exponent digits
static std::string py_repr(double v) {
  if (std::isnan(v)) return "nan";
  if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
  if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
  std::string s(buf, r.ptr);                                 // [-]d[.ddd]e[+-]XX
  std::string sign;
  if (s[0] == '-') { sign = "-"; s = s.substr(1); }
  const size_t e = s.find('e');
  std::string mant = s.substr(0, e);
  const int ex = atoi(s.c_str() + e + 1);
  std::string digits;
  for (char c : mant) if (c != '.') digits += c;
  const int decpt = ex + 1;                                  // value = 0.digits x 10^decpt
  std::string out;
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) out = "0." + std::string((size_t)(-decpt), '0') + digits;
    else if ((size_t)decpt >= digits.size()) out = digits + std::string((size_t)decpt - digits.size(), '0') + ".0";
    else out = digits.substr(0, (size_t)decpt) + "." + digits.substr((size_t)decpt);
  } else {
    out = digits.substr(0, 1);
    if (digits.size() > 1) out += "." + digits.substr(1);
    char eb[16];
    snprintf(eb, sizeof eb, "e%c%02d", ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
    out += eb;
  }
  return sign + out;
}

struct Edge { int other; int ov; double cc; double norm; };
static inline bool edge_eq(const Edge& a, const Edge& b) { return a.other == b.other && a.ov == b.ov && a.cc == b.cc && a.norm == b.norm; }

struct Node {
  const char* str; int slen;          // bases (S: "Start_", E: "_End")
  double weight; int L;
  int key;                            // the original node id a clone descends from (S / E: -1)
  bool orig;
  std::vector<Edge> ine, oute;
};

struct Req {                          // one path_decompose call that needs LP trials (path_decompose_sparse.py:33-100)
  int m, n, trials, sparsity;
  uint64_t pid;
  std::vector<double> a, b, p, a_s, b_s;
  double tol, scale;
};

// one component = one algorithm_SF.py process (run_MB_SF_fn.py:242-250)
struct Component {
  std::vector<Node> nodes;            // node pool; S = nodes[nS], E = nodes[nE]
  std::vector<int> all;               // `allnodes`
  int S = -1, E = -1;
  std::vector<std::vector<int>> known;           // known paths as node indices
  std::vector<std::vector<int>> pfn;             // per original node: first PATH_SPARSITY paths through it
  int comp_id = 0, n_dec = 0;
  // state of the running algorithm2 pass
  size_t idx = 0;
  bool started = false, finished = false;
  // the decomposition waiting for its LP answer
  int cur = -1;
  std::vector<int> inn, outn;
  std::vector<Edge> in_attr, out_attr;
  Req req;
  std::vector<std::vector<double>> answer;
  std::string fasta;                  // records of this component
};

static const char kStart[] = "Start_";
static const char kEnd[] = "_End";

static int n_trials(int m, int n) { const long v = 2L * m * n * std::max(m, n); return (int)std::min<long>(v, 100); }

// path_decompose up to the trial loop; returns true when the answer is closed-form (in c.answer)
static bool prepare(Component& c, const std::vector<double>& a0, const std::vector<double>& b0, const std::vector<std::vector<int>>& P, uint64_t pid) {
  const int m = (int)a0.size(), n = (int)b0.size();
  c.answer.clear();
  if (m == 0 || n == 0) return true;
  if (m == 1) { c.answer.assign(1, b0); return true; }
  if (n == 1) { for (double v : a0) c.answer.push_back(std::vector<double>(1, v)); return true; }
  double sa = 0.0, sb = 0.0;
  for (double v : a0) sa += v;
  for (double v : b0) sb += v;
  if (sa <= 0 || sb <= 0) { c.answer.assign(m, std::vector<double>(n, 0.0)); return true; }
  std::vector<double> a = a0, b = b0;
  if (sa > sb) { const double cst = sa - sb; for (double& k : b) k = k + cst * k / sb; }
  else { const double cst = sb - sa; for (double& k : a) k = k + cst * k / sa; }
  Req& q = c.req;
  q.m = m; q.n = n; q.a = a; q.b = b; q.pid = pid; q.sparsity = PATH_SPARSITY;
  q.p.assign((size_t)m * n, 0.0);
  for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) q.p[(size_t)j * m + i] = 1.0 - (double)P[i][j];
  std::vector<double> rhs(a);
  rhs.insert(rhs.end(), b.begin(), b.end());
  rhs.resize((size_t)m + n - 1);
  double weight = 0.0;
  for (double v : a) weight += std::fabs(v);
  q.tol = 0.001 * weight;
  double mx = rhs[0];
  for (double v : rhs) mx = std::max(mx, v);
  q.scale = std::max(mx, 1e-100) * 0.01;
  q.a_s.clear(); q.b_s.clear();
  for (int i = 0; i < m; i++) q.a_s.push_back(rhs[i] / q.scale);
  for (int j = m; j < m + n - 1; j++) q.b_s.push_back(rhs[j] / q.scale);
  double tot = 0.0;
  for (double v : q.a_s) tot += v;
  for (double v : q.b_s) tot -= v;
  q.b_s.push_back(tot > 0 ? tot : 0.0);
  q.trials = n_trials(m, n);
  return false;
}

// path_decompose after the LP solves (path_decompose_sparse.py:118-192).  xs: [cell j*m+i][trial], scaled solutions
static void finish(Component& c, const double* xs) {
  const Req& q = c.req;
  const int m = q.m, n = q.n, mn = m * n, T = q.trials;
  std::vector<double> thr(mn);
  for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) thr[(size_t)j * m + i] = 0.4 * std::min(q.a[i], q.b[j]);
  std::vector<double> temp((size_t)mn), cur_ans;
  int curr_min = mn + 1, curr_mult = 0;
  double curr_on = 0.0;
  bool have = false;
  for (int t = 0; t < T; t++) {
    int s = 0;
    for (int k = 0; k < mn; k++) {
      double v = xs[(size_t)k * T + t] * q.scale;
      if (v < thr[k] || v < q.tol || v < 0) v = 0.0;
      temp[k] = v;
      if (v != 0 && q.p[k] > 0) s++;
    }
    if (s > curr_min) continue;
    double dot = 0.0;
    for (int k = 0; k < mn; k++) dot += q.p[k] * temp[k];
    if (s < curr_min) { curr_min = s; cur_ans = temp; curr_mult = 0; curr_on = dot; have = true; }
    else {
      double d2 = 0.0;
      for (int k = 0; k < mn; k++) { const double d = cur_ans[k] - temp[k]; d2 += d * d; }
      if (std::sqrt(d2) > q.tol) curr_mult++;
      double st = 0.0, sc = 0.0;
      for (double v : temp) st += v;
      for (double v : cur_ans) sc += v;
      if ((std::fabs(st - sc) < q.tol && dot < curr_on) || st > sc) { cur_ans = temp; curr_on = dot; }
    }
  }
  (void)have;
  c.answer.assign(m, std::vector<double>(n, 0.0));
  for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) c.answer[i][j] = cur_ans[(size_t)j * m + i];
  if (q.sparsity && mn > q.sparsity) {
    // sorted(cells, key=value)[::-1][:sparsity]: stable ascending, reversed
    std::vector<std::pair<int, int>> cells;
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) cells.push_back({i, j});
    std::stable_sort(cells.begin(), cells.end(), [&](const std::pair<int, int>& x, const std::pair<int, int>& y) { return c.answer[x.first][x.second] < c.answer[y.first][y.second]; });
    std::reverse(cells.begin(), cells.end());
    std::vector<std::vector<double>> nw(m, std::vector<double>(n, 0.0));
    for (int k = 0; k < q.sparsity && k < (int)cells.size(); k++) nw[cells[k].first][cells[k].second] = c.answer[cells[k].first][cells[k].second];
    c.answer.swap(nw);
  }
}

static void add_E(Component& c, int x) {
  Node& nd = c.nodes[x];
  nd.oute.push_back(Edge{c.E, 0, nd.weight, 0});
  c.nodes[c.E].ine.push_back(Edge{x, 0, nd.weight, 0});
  c.nodes[c.E].weight += nd.weight;
}

// the clones of the decomposed node + its removal (algorithm_SF.py:515-547), then the pass goes on
static void apply_flow(Component& c) {
  const int node = c.cur;
  const int m = (int)c.inn.size(), n = (int)c.outn.size();
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      const double cc = c.answer[i][j];
      if (cc != 0) {
        const Edge ia = c.in_attr[i], oa = c.out_attr[j];
        Node nn;
        nn.str = c.nodes[node].str; nn.slen = c.nodes[node].slen; nn.weight = cc; nn.L = c.nodes[node].L; nn.key = c.nodes[node].key; nn.orig = false;
        const int id = (int)c.nodes.size();
        c.nodes.push_back(nn);
        c.nodes[id].ine.push_back(Edge{c.inn[i], ia.ov, cc, ia.norm});
        c.nodes[c.inn[i]].oute.push_back(Edge{id, ia.ov, cc, ia.norm});
        c.nodes[id].oute.push_back(Edge{c.outn[j], oa.ov, cc, oa.norm});
        c.nodes[c.outn[j]].ine.push_back(Edge{id, oa.ov, cc, oa.norm});
        c.all.push_back(id);
      }
    }
  // :532-543 -- removal while iterating, as written: Python's list iterator skips the element after a removed one
  for (size_t ei = 0; ei < c.nodes[node].ine.size(); ei++) {
    std::vector<Edge>& lst = c.nodes[c.nodes[node].ine[ei].other].oute;
    size_t it = 0;
    while (it < lst.size()) {
      const Edge oe = lst[it];
      it++;
      if (oe.other == node) { for (size_t q = 0; q < lst.size(); q++) if (edge_eq(lst[q], oe)) { lst.erase(lst.begin() + q); break; } }
    }
  }
  for (size_t ei = 0; ei < c.nodes[node].oute.size(); ei++) {
    std::vector<Edge>& lst = c.nodes[c.nodes[node].oute[ei].other].ine;
    size_t it = 0;
    while (it < lst.size()) {
      const Edge ie = lst[it];
      it++;
      if (ie.other == node) { for (size_t q = 0; q < lst.size(); q++) if (edge_eq(lst[q], ie)) { lst.erase(lst.begin() + q); break; } }
    }
  }
  for (size_t k = 0; k < c.all.size(); k++) if (c.all[k] == node) { c.all.erase(c.all.begin() + k); break; }     // (the pass then skips one element: quirk 21)
  c.cur = -1;
}

static bool same_str(const Node& a, const Node& b) { return a.slen == b.slen && memcmp(a.str, b.str, (size_t)a.slen) == 0; }

// runs algorithm2 until a decomposition needs LP trials (returns true, request in c.req) or the component is done (false)
static bool advance(Component& c) {
  if (c.finished) return false;
  if (!c.started) { c.started = true; c.idx = 0; if (c.all.size() <= 3) { c.finished = true; return false; } }
  while (true) {
    while (c.idx < c.all.size()) {
      const int node = c.all[c.idx];
      c.idx++;
      if (node == c.S || node == c.E || c.nodes[node].ine.size() <= 1) continue;
      if (c.nodes[node].oute.empty()) add_E(c, node);
      const Node& nd = c.nodes[node];
      c.inn.clear(); c.outn.clear(); c.in_attr.clear(); c.out_attr.clear();
      std::vector<double> a, b;
      for (const Edge& e : nd.ine) { c.inn.push_back(e.other); a.push_back(e.cc); }
      for (const Edge& e : nd.oute) { c.outn.push_back(e.other); b.push_back(e.cc); }
      // in_attr = {id(e[0]): [e[1], e[3]]}: a dict -- for a neighbour that occurs twice the LAST edge's attributes win
      for (size_t i = 0; i < c.inn.size(); i++) { Edge at = nd.ine[i]; for (size_t q = 0; q < nd.ine.size(); q++) if (nd.ine[q].other == c.inn[i]) at = nd.ine[q]; c.in_attr.push_back(at); }
      for (size_t j = 0; j < c.outn.size(); j++) { Edge at = nd.oute[j]; for (size_t q = 0; q < nd.oute.size(); q++) if (nd.oute[q].other == c.outn[j]) at = nd.oute[q]; c.out_attr.push_back(at); }
      const int m = (int)a.size(), n = (int)b.size();
      std::vector<std::vector<int>> P(m, std::vector<int>(n, 0));
      if (nd.orig && !c.pfn[node].empty()) {                 // support matrix :438-498 (singleton constituents)
        for (int mi = 0; mi < m; mi++) {
          const int u = c.inn[mi];
          if (!c.nodes[u].orig) continue;
          for (int ni = 0; ni < n; ni++) {
            const int w = c.outn[ni];
            if (!c.nodes[w].orig) continue;
            for (int cp : c.pfn[node]) {
              if (std::find(c.pfn[u].begin(), c.pfn[u].end(), cp) == c.pfn[u].end()) continue;
              if (std::find(c.pfn[w].begin(), c.pfn[w].end(), cp) == c.pfn[w].end()) continue;
              const std::vector<int>& nl = c.known[cp];
              size_t k = 0;
              while (k < nl.size() && nl[k] != node) k++;
              if (k == nl.size()) continue;
              const bool lg = k == 0 || same_str(c.nodes[nl[k - 1]], c.nodes[u]);
              const bool rg = k + 1 == nl.size() || same_str(c.nodes[nl[k + 1]], c.nodes[w]);
              if (lg && rg) P[mi][ni] = 1;
            }
          }
        }
      }
      c.cur = node;
      const bool done = prepare(c, a, b, P, ((uint64_t)c.comp_id << 20) + (uint64_t)c.n_dec);
      c.n_dec++;
      if (!done) return true;
      apply_flow(c);
    }
    // reducible() :357-370
    bool red = false;
    for (int x : c.all) if (x != c.S && x != c.E && !c.nodes[x].ine.empty() && !c.nodes[x].oute.empty() && c.nodes[x].ine.size() > 1) { red = true; break; }
    if (!red) { c.finished = true; return false; }
    std::vector<int> rest;
    for (int x : c.all) if (x != c.S && x != c.E) rest.push_back(x);
    std::stable_sort(rest.begin(), rest.end(), [&](int x, int y) { return c.nodes[x].key < c.nodes[y].key; });
    c.all.clear();
    c.all.push_back(c.S);
    c.all.insert(c.all.end(), rest.begin(), rest.end());
    c.all.push_back(c.E);
    c.idx = 0;
  }
}

// read_Y_paths :564-613 + the header format of :608-609
static void emit(Component& c, const std::string& sname, int comp) {
  struct Frame { int node; size_t edge; size_t cur_len, names_len; double sw; long sn; };
  std::string cur, names;
  std::vector<Frame> st;
  int n_out = 0;
  // iterative depth-first walk; `sw` starts as the int 0 and becomes a float with the first node weight added
  std::vector<std::pair<std::string, std::pair<std::string, std::string>>> out;
  auto enter = [&](int node, int overlap, double sw, long sn) -> bool {
    const Node& nd = c.nodes[node];
    const size_t cl = cur.size(), nl = names.size();
    if (overlap < nd.slen) cur.append(nd.str + overlap, (size_t)(nd.slen - overlap));
    names += "->";
    names += node == c.S ? "S" : node == c.E ? "E" : std::to_string(nd.key);
    if (nd.oute.empty()) {
      if (cur.size() >= 4 && cur.compare(cur.size() - 4, 4, "_End") == 0) {
        const std::string s = cur.substr(0, cur.size() - 4);
        const std::string w = sn > 0 ? py_repr(sw / (double)sn) : std::string("0");
        if (s.size() > 6) {
          c.fasta += ">Shannon_" + sname + " " + std::to_string(comp) + "_" + std::to_string(n_out) + "\t" + w + "\t" + names + "\n" + s.substr(6) + "\n";
        }
        n_out++;
      }
      cur.resize(cl); names.resize(nl);
      return false;
    }
    st.push_back(Frame{node, 0, cl, nl, sw + nd.weight, sn + nd.L});
    return true;
  };
  enter(c.S, 0, 0.0, 0);
  while (!st.empty()) {
    Frame& f = st.back();
    const Node& nd = c.nodes[f.node];
    if (f.edge >= nd.oute.size()) { cur.resize(f.cur_len); names.resize(f.names_len); st.pop_back(); continue; }
    const Edge e = nd.oute[f.edge++];
    const double sw = f.sw; const long sn = f.sn;
    enter(e.other, e.ov, sw, sn);
  }
}

}  // namespace

// The host threads of one shn_sparse_flow call, kept for the whole call: a round advances a few hundred components in microseconds
// each, and until round 6 every phase of every round started and joined its own std::threads -- 2 x 16 thread starts per round, 890
// rounds per step at bench.py --config 2p: 4.7 of the stage's 8.4 s.  run(n, chunk, fn): fn(i) for every i < n, chunks of `chunk`
// handed out by an atomic counter, the caller works too and returns when every index is done.  The workers spin for a few
// microseconds between jobs (the next phase of a round follows at once) before they sleep on the condition variable.
struct SflowPool {
  // Every worker waits on a word of its own (a futex: no mutex to take back after the wake-up, which is what serialised the 15
  // workers of a phase behind one another) and a phase wakes only the workers it wants: `run` may ask for fewer threads than the
  // pool holds -- a wave of partitions started beside 15 graph stages keeps to two threads then and spreads when they are done.
  struct alignas(64) Slot { std::atomic<uint32_t> gen{0}; std::atomic<uint32_t> asleep{0}; };
  std::vector<std::thread> th;
  std::unique_ptr<Slot[]> slot;
  std::atomic<size_t> next{0};
  std::atomic<int> busy{0};
  std::atomic<bool> stop{false};
  std::atomic<uint32_t> spin_us{20};
  size_t n = 0, chunk = 1;
  const std::function<void(size_t)>* fn = nullptr;
  static void futex_wait(std::atomic<uint32_t>* w, uint32_t v) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAIT_PRIVATE, v, nullptr, nullptr, 0); }
  static void futex_wake(std::atomic<uint32_t>* w) { syscall(SYS_futex, (uint32_t*)w, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }
  explicit SflowPool(unsigned workers) : slot(new Slot[std::max(1u, workers)]) {
    for (unsigned t = 0; t < workers; t++) th.emplace_back([this, t]() { loop(t); });
  }
  ~SflowPool() {
    stop.store(true);
    for (size_t t = 0; t < th.size(); t++) { slot[t].gen.fetch_add(1); futex_wake(&slot[t].gen); }
    for (auto& x : th) x.join();
  }
  void drain() {
    while (true) {
      const size_t i0 = next.fetch_add(chunk);
      if (i0 >= n) break;
      const size_t i1 = std::min(n, i0 + chunk);
      for (size_t i = i0; i < i1; i++) (*fn)(i);
    }
  }
  void loop(unsigned t) {
    Slot& me = slot[t];
    uint32_t seen = 0;
    auto clock_us = []() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (true) {
      // spin for as long as the phases of a round are apart (the caller says how long that is worth: spin_us), then sleep
      bool got = me.gen.load(std::memory_order_acquire) != seen;
      if (!got) {
        const uint64_t t0 = clock_us(), lim = spin_us.load(std::memory_order_relaxed);
        while (!got) {
          for (int sp = 0; sp < 64 && !got; sp++) { got = me.gen.load(std::memory_order_acquire) != seen; if (!got) __builtin_ia32_pause(); }
          if (!got && clock_us() - t0 >= lim) break;
        }
      }
      while (!got) {
        me.asleep.store(1);                                         // (seq_cst, as is the publisher's gen++: one of the two sees the other)
        if (me.gen.load() == seen) futex_wait(&me.gen, seen);
        me.asleep.store(0);
        got = me.gen.load(std::memory_order_acquire) != seen;
      }
      if (stop.load()) return;
      seen = me.gen.load(std::memory_order_acquire);                // (a job: n / chunk / fn written, busy and next set, THEN the wanted workers' gen++)
      drain();
      busy.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  // `threads`: the calling thread included; at most the pool's workers + 1
  void run(size_t n_items, size_t chunk_, const std::function<void(size_t)>& f, unsigned threads = ~0u) {
    if (!n_items) return;
    const size_t want = std::min<size_t>(th.size(), threads > 0 ? threads - 1 : 0);
    if (!want || n_items <= chunk_) { for (size_t i = 0; i < n_items; i++) f(i); return; }
    n = n_items; chunk = std::max<size_t>(1, chunk_); fn = &f;
    next.store(0); busy.store((int)want + 1);
    for (size_t t = 0; t < want; t++) { slot[t].gen.fetch_add(1); if (slot[t].asleep.load()) futex_wake(&slot[t].gen); }
    drain();
    busy.fetch_sub(1, std::memory_order_acq_rel);
    while (busy.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();      // (the stragglers are inside their last chunk)
  }
};

struct shn_sflow { std::vector<std::string> text; };

extern "C" void shn_sflow_destroy(shn_sflow* s) { delete s; }
extern "C" uint64_t shn_sflow_text_size(const shn_sflow* s, uint32_t g) { return (s && g < s->text.size()) ? s->text[g].size() : 0; }
extern "C" int shn_sflow_text(const shn_sflow* s, uint32_t g, uint8_t* out) {
  if (!s || g >= s->text.size() || !out) return shn_fail(SHN_ERR_ARG, "shn_sflow_text: bad argument");
  std::copy(s->text[g].begin(), s->text[g].end(), (char*)out);
  return SHN_OK;
}

// graphs[g]: the multibridged graph of partition g (shn_mbgraph_run*); snames[g]: "<sample>_<partition>" (NUL-terminated).
// Text g = the reconstructed FASTA of partition g: the records of its components in order (component c uses the LP problem
// ids (c << 20) + call number), then its single nodes (single_nodes_to_fasta, algorithm_SF.py:74-88).
extern "C" int shn_sparse_flow(shn_ctx* ctx, const shn_graph* const* graphs, uint32_t n_graphs, const char* const* snames, uint64_t seed, shn_sflow** out);
int shn_graph_partitions_running();                  // mbgraph_host.hip
static thread_local bool t_sflow_beside = false;     // this thread is one of several inside shn_sparse_flow: no helper threads of its own for small inputs
// One of several calls made by host threads at the same time (a partition's components right behind its graph stage, while other
// partitions are still at theirs): the LP batches run on the calling thread's own stream and workspaces (shn_thread_ctx).  The
// answers are those of one call over all graphs: a component's random costs depend on its index inside its own graph only.
extern "C" int shn_sparse_flow_thread(shn_ctx* ctx, const shn_graph* const* graphs, uint32_t n_graphs, const char* const* snames, uint64_t seed, shn_sflow** out) {
  if (!ctx) return shn_fail(SHN_ERR_ARG, "shn_sparse_flow_thread: ctx is NULL");
  struct Flag { Flag() { t_sflow_beside = true; } ~Flag() { t_sflow_beside = false; } } flag;
  return shn_sparse_flow(shn_thread_ctx(ctx), graphs, n_graphs, snames, seed, out);
}
extern "C" int shn_sparse_flow(shn_ctx* ctx, const shn_graph* const* graphs, uint32_t n_graphs, const char* const* snames, uint64_t seed, shn_sflow** out) {
  if (!ctx || !out || (n_graphs && (!graphs || !snames))) return shn_fail(SHN_ERR_ARG, "shn_sparse_flow: NULL argument");
  *out = nullptr;
  const bool dbg = getenv("SHN_DEBUG") != nullptr || getenv("SHN_SFLOW_LAPS") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = now(), t_adv = 0, t_lp = 0, t_fin = 0, t_pack = 0;
  size_t n_small_rounds = 0, n_pend_total = 0; uint64_t n_trials_total = 0;
  uint64_t h_prob[8] = {0}, h_roundmax[8] = {0};          // problems / rounds by the size class of (the largest) max(m, n): <=2, 3, 4, 5-6, 7-8, 9-11, 12-16, more
  auto lap = [&](const char* what) { if (dbg) { const double t = now(); fprintf(stderr, "[sflow] %-28s %8.3f s\n", what, t - t_lap); t_lap = t; } };
  std::vector<Component> comps;
  std::vector<uint32_t> comp_graph;
  std::vector<int> comp_index;
  for (uint32_t g = 0; g < n_graphs; g++) {
    const shn_graph* G = graphs[g];
    if (!G) return shn_fail(SHN_ERR_ARG, "shn_sparse_flow: NULL graph");
    const size_t nc = G->comp_node_off.size() - 1;
    for (size_t ci = 0; ci < nc; ci++) {
      comps.emplace_back();
      comp_graph.push_back(g);
      comp_index.push_back((int)ci);
    }
  }
  // build the components (ParseNodeFile / ParseEdgeFile / ParseKnownPathsFile / findStartAndEnd2) on host threads
  auto build = [&](size_t k) {
    Component& c = comps[k];
    const shn_graph* G = graphs[comp_graph[k]];
    const int ci = comp_index[k];
    c.comp_id = ci;
    const uint64_t a = G->comp_node_off[ci], b = G->comp_node_off[ci + 1];
    const int nn = (int)(b - a);
    c.nodes.reserve((size_t)nn * 2 + 8);
    for (int i = 0; i < nn; i++) {
      Node nd;
      nd.str = G->n_bases.data() + G->n_off[a + i]; nd.slen = (int)(G->n_off[a + i + 1] - G->n_off[a + i]);
      nd.weight = G->n_cc_int[a + i] ? 0.0 : G->n_cc[a + i];
      nd.L = nd.slen; nd.key = i; nd.orig = true;
      c.nodes.push_back(nd);
      c.all.push_back(i);
    }
    for (uint64_t e = G->comp_edge_off[ci]; e < G->comp_edge_off[ci + 1]; e++) {
      const int s = G->e_in[e], d = G->e_out[e];
      c.nodes[s].oute.push_back(Edge{d, G->e_w[e], G->e_cc[e], (double)(int)G->e_norm[e]});
      c.nodes[d].ine.push_back(Edge{s, G->e_w[e], G->e_cc[e], (double)(int)G->e_norm[e]});
    }
    c.pfn.assign((size_t)nn, std::vector<int>());
    int pi = 0;
    for (uint64_t p = G->comp_path_off[ci]; p < G->comp_path_off[ci + 1]; p++, pi++) {
      std::vector<int> lst;
      for (uint64_t q = G->p_off[p]; q < G->p_off[p + 1]; q++) lst.push_back(G->p_ids[q]);
      for (int x : lst) if ((int)c.pfn[x].size() < PATH_SPARSITY) c.pfn[x].push_back(pi);
      c.known.push_back(lst);
    }
    Node S, E;
    S.str = kStart; S.slen = 6; S.weight = 0; S.L = 0; S.key = -1; S.orig = false;
    E.str = kEnd; E.slen = 4; E.weight = 0; E.L = 0; E.key = -1; E.orig = false;
    c.S = (int)c.nodes.size(); c.nodes.push_back(S);
    c.E = (int)c.nodes.size(); c.nodes.push_back(E);
    for (int i = 0; i < nn; i++) {
      if (c.nodes[i].ine.empty()) {
        c.nodes[i].ine.push_back(Edge{c.S, 0, c.nodes[i].weight, 0});
        c.nodes[c.S].oute.push_back(Edge{i, 0, c.nodes[i].weight, 0});
        c.nodes[c.S].weight += c.nodes[i].weight;
      }
      if (c.nodes[i].oute.empty()) add_E(c, i);
    }
    c.all.push_back(c.S);
    c.all.push_back(c.E);
    c.pfn.resize(c.nodes.size());
  };
  // (a call beside other partitions' graph stages keeps to its own thread unless it is large: the machine is busy)
  size_t n_nodes_all = 0;
  for (uint32_t g = 0; g < n_graphs; g++) n_nodes_all += graphs[g]->n_off.size();
  // (beside the graph stage the cores are taken: a few threads for a wave of partitions, the calling thread alone for one partition)
  const bool small_job = t_sflow_beside ? (comps.size() < 1024 && n_nodes_all < (1u << 15)) : (comps.size() < 256 && n_nodes_all < (1u << 16));
  const unsigned cpus = (unsigned)std::max(1, shn_host_cpus());
  // The threads of a phase are decided when it starts: the cores the graph stages leave free right now, shared among the calls
  // that are in here together (waves of partitions beside the graph stage), two at least.  A call that has the machine to itself
  // keeps its workers spinning between the phases of a round (they are ~0.3 ms apart: one LP batch on the device); beside others
  // they spin for 20 us and sleep.
  static std::atomic<int> calls_with_pool{0};
  struct InHere { bool on; explicit InHere(bool b) : on(b) { if (on) calls_with_pool.fetch_add(1); } ~InHere() { if (on) calls_with_pool.fetch_sub(1); } } in_here(!small_job);
  const bool beside = t_sflow_beside;
  auto threads_now = [&]() -> unsigned {
    if (small_job) return 1u;
    const unsigned busy = (unsigned)std::max(0, shn_graph_partitions_running());      // partitions at their graphs right now: a core each
    const unsigned calls = (unsigned)std::max(1, calls_with_pool.load());
    const unsigned free_ = cpus > busy ? cpus - busy : 0u;
    return beside ? std::max(2u, std::min(32u, free_ / calls)) : std::max(1u, std::min(32u, cpus));
  };
  static const unsigned spin_alone = getenv("SHN_SFLOW_SPIN_US") ? (unsigned)atoi(getenv("SHN_SFLOW_SPIN_US")) : 400u;
  const unsigned nt_max = small_job ? 1u : std::max(2u, std::min(32u, cpus));
  unsigned nt = threads_now();
  SflowPool pool(nt_max > 1 ? nt_max - 1 : 0);
  auto retune = [&]() {
    nt = threads_now();
    pool.spin_us.store(calls_with_pool.load() <= 1 && shn_graph_partitions_running() <= 0 ? spin_alone : 20u, std::memory_order_relaxed);
  };
  retune();
  // chunks: a few per thread, so that a handful of large components (the repeat-linked families of --config 2p: 610 components of
  // ~3,000 nodes) do not end up on one thread behind 63 others
  auto chunk_of = [&](size_t n_items) { return std::max<size_t>(1, std::min<size_t>(64, n_items / ((size_t)nt * 8))); };
  auto parallel = [&](const std::function<void(size_t)>& fn) { retune(); pool.run(comps.size(), chunk_of(comps.size()), fn, nt); };
  parallel(build);
  lap("build components");
  int n_round = 0;
  std::atomic<uint64_t> busy_adv_ns{0};
  const double t_rounds0 = now();
  unsigned nt_lo = nt, nt_hi = nt;
  // rounds: every component runs to its next decomposition that needs LP trials; all of them go to the device in one batch
  std::vector<size_t> active(comps.size());
  for (size_t k = 0; k < comps.size(); k++) active[k] = k;
  std::vector<uint8_t> wants(comps.size(), 0);
  while (!active.empty()) {
    double tr0 = now();
    n_round++;
    if ((n_round & 15) == 1) { retune(); nt_lo = std::min(nt_lo, nt); nt_hi = std::max(nt_hi, nt); }
    // advance the active components (independent of each other)
    pool.run(active.size(), std::max<size_t>(1, std::min<size_t>(16, active.size() / ((size_t)nt * 4))),
             [&](size_t i) {
               if (!dbg) { wants[active[i]] = advance(comps[active[i]]) ? 1 : 0; return; }
               const double ta = now();
               wants[active[i]] = advance(comps[active[i]]) ? 1 : 0;
               busy_adv_ns.fetch_add((uint64_t)((now() - ta) * 1e9), std::memory_order_relaxed);
             }, nt);
    std::vector<size_t> pend;
    for (size_t k : active) if (wants[k]) pend.push_back(k);
    t_adv += now() - tr0; tr0 = now();
    if (pend.empty()) break;
    n_pend_total += pend.size(); if (pend.size() < 64) n_small_rounds++;
    std::vector<uint32_t> m(pend.size()), n(pend.size()), tr(pend.size());
    std::vector<uint64_t> pid(pend.size());
    std::vector<double> ab;
    std::vector<uint8_t> mask;
    std::vector<uint64_t> ooff(pend.size() + 1, 0);
    for (size_t i = 0; i < pend.size(); i++) {
      const Req& q = comps[pend[i]].req;
      m[i] = (uint32_t)q.m; n[i] = (uint32_t)q.n; tr[i] = (uint32_t)q.trials; pid[i] = q.pid;
      ab.insert(ab.end(), q.a_s.begin(), q.a_s.end());
      ab.insert(ab.end(), q.b_s.begin(), q.b_s.end());
      for (double v : q.p) mask.push_back(v > 0 ? 1 : 0);
      ooff[i + 1] = ooff[i] + (uint64_t)q.m * q.n * q.trials;
    }
    std::vector<double> flows(ooff.back());
    for (size_t i = 0; i < pend.size(); i++) n_trials_total += tr[i];
    if (dbg) {
      auto cls = [](uint32_t x) { return x <= 2 ? 0 : x == 3 ? 1 : x == 4 ? 2 : x <= 6 ? 3 : x <= 8 ? 4 : x <= 11 ? 5 : x <= 16 ? 6 : 7; };
      int mx = 0;
      for (size_t i = 0; i < pend.size(); i++) { const int c_ = cls(std::max(m[i], n[i])); h_prob[c_]++; mx = std::max(mx, c_); }
      h_roundmax[mx]++;
    }
    t_pack += now() - tr0; tr0 = now();
    int rc = shn_lp_solve_batch(ctx, (uint32_t)pend.size(), m.data(), n.data(), tr.data(), pid.data(), ab.data(), mask.data(), seed, flows.data());
    if (rc) return rc;
    t_lp += now() - tr0; tr0 = now();
    pool.run(pend.size(), std::max<size_t>(1, std::min<size_t>(16, pend.size() / ((size_t)nt * 4))),
             [&](size_t i) { Component& c = comps[pend[i]]; finish(c, flows.data() + ooff[i]); apply_flow(c); }, nt);
    active.swap(pend);
    t_fin += now() - tr0;
  }
  if (dbg) fprintf(stderr, "[sflow] %zu components of %u graphs, %d rounds (%zu of them with fewer than 64 problems; %zu problems, %llu trials in all): advance %.3f s, pack %.3f s, "
                   "LP batches %.3f s, finish+apply %.3f s; rounds from %.3f to %.3f (clock mod 100 s), threads %u..%u; inside advance %.3f thread-s\n", comps.size(), n_graphs, n_round, n_small_rounds, n_pend_total, (unsigned long long)n_trials_total, t_adv, t_pack, t_lp, t_fin,
                   std::fmod(t_rounds0, 100.0), std::fmod(now(), 100.0), nt_lo, nt_hi, (double)busy_adv_ns.load() * 1e-9);
  if (dbg && n_pend_total > 1000) fprintf(stderr, "[sflow]   problems by max(m, n) <=2 / 3 / 4 / 5-6 / 7-8 / 9-11 / 12-16 / more: %llu %llu %llu %llu %llu %llu %llu %llu; rounds by their largest problem: %llu %llu %llu %llu %llu %llu %llu %llu\n",
                   (unsigned long long)h_prob[0], (unsigned long long)h_prob[1], (unsigned long long)h_prob[2], (unsigned long long)h_prob[3], (unsigned long long)h_prob[4], (unsigned long long)h_prob[5], (unsigned long long)h_prob[6], (unsigned long long)h_prob[7],
                   (unsigned long long)h_roundmax[0], (unsigned long long)h_roundmax[1], (unsigned long long)h_roundmax[2], (unsigned long long)h_roundmax[3], (unsigned long long)h_roundmax[4], (unsigned long long)h_roundmax[5], (unsigned long long)h_roundmax[6], (unsigned long long)h_roundmax[7]);
  t_lap = now();
  // transcripts of every component, then the partition texts
  {
    std::vector<std::string> names(n_graphs);
    for (uint32_t g = 0; g < n_graphs; g++) names[g] = snames[g] ? snames[g] : "";
    parallel([&](size_t k) { emit(comps[k], names[comp_graph[k]], comp_index[k]); });
  }
  lap("emit");
  shn_sflow* R = new shn_sflow();
  R->text.assign(n_graphs, std::string());
  // the text of a partition: its components' transcripts, then its single nodes -- partitions side by side on the host threads
  std::vector<size_t> first_comp(n_graphs + 1, comps.size());
  for (size_t k = comps.size(); k-- > 0;) first_comp[comp_graph[k]] = k;
  for (uint32_t g = n_graphs; g-- > 0;) if (first_comp[g] == comps.size()) first_comp[g] = first_comp[g + 1];
  auto text_of = [&](uint32_t g) {
    // single_nodes_to_fasta -- including its quirk of not skipping the header line of single_nodes.txt
    const shn_graph* G = graphs[g];
    const std::string sname = snames[g] ? snames[g] : "";
    std::string& t = R->text[g];
    size_t bytes = 64 + sname.size();
    for (size_t k = first_comp[g]; k < first_comp[g + 1]; k++) bytes += comps[k].fasta.size();
    bytes += G->s_bases.size() + (G->s_off.size()) * (48 + sname.size());
    t.reserve(bytes);
    for (size_t k = first_comp[g]; k < first_comp[g + 1]; k++) t += comps[k].fasta;
    t += ">Shannon_" + sname + "_single_0\t Copycount:Copycount\nBases\n";
    for (size_t i = 0; i + 1 < G->s_off.size(); i++) {
      const double cc = G->s_cc[i];
      t += ">Shannon_" + sname + "_single_" + std::to_string(i + 1) + "\t Copycount:" + (cc == 0 ? std::string("0") : py_repr(cc)) + "\n";
      t.append(G->s_bases, G->s_off[i], G->s_off[i + 1] - G->s_off[i]);
      t += "\n";
    }
  };
  retune();
  pool.run(n_graphs, 1, [&](size_t g) { text_of((uint32_t)g); }, nt);
  lap("partition texts");
  // The components hold ~10^2 small vectors each (26 ms to give back on 16 threads at BASELINE configs[2]): they go to a
  // background thread, which frees them while the caller merges the transcripts; the thread of the call before is joined first, the
  // last one when the library is unloaded.  SHN_SFLOW_FREE_NOW=1: on the host threads, before returning.
  if (getenv("SHN_SFLOW_FREE_NOW")) parallel([&](size_t k) { Component gone; std::swap(gone, comps[k]); });
  else {
    struct Reaper {
      std::mutex mu; std::thread t;
      ~Reaper() { if (t.joinable()) t.join(); }
      void take(std::vector<Component>* v) {
        std::lock_guard<std::mutex> lk(mu);
        if (t.joinable()) t.join();
        t = std::thread([v]() { delete v; });
      }
    };
    static Reaper reaper;
    reaper.take(new std::vector<Component>(std::move(comps)));
  }
  lap("free components");
  *out = R;
  return SHN_OK;
}
