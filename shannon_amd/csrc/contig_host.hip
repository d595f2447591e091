// Host-side (C++) contig bookkeeping of extension_correction.run_correction, rows a5-a6:
// duplicate_check (extension_correction.py:247-270) and the contig graph by shared K-mers
// (:372-397), sequential over the candidate contigs in seed order exactly as the reference.
// This is control logic over contig bases (the k-mer-level walks are on the GPU, extend.hip); it is
// native code because at 10M reads ~2,000 candidate contigs x ~2,000 bases of Python dict work
// were a third of the host time.
#include "common.h"
#include "flatmap.h"
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <cstring>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <cstdio>

static inline int code_of(uint8_t c) {
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// packed keys of all k-windows of s[0..L) (assumes ACGT only)
static void window_keys(const uint8_t* s, uint32_t L, int k, std::vector<uint64_t>& out) {
  out.clear();
  if ((int)L < k) return;
  uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1), v = 0;
  for (uint32_t i = 0; i < L; i++) {
    v = ((v << 2) | (uint64_t)(code_of(s[i]) & 3)) & mask;
    if ((int)i >= k - 1) out.push_back(v);
  }
}

// per candidate of the last shn_contig_graph call of this thread: the hit count of its `best` contig (0: no hit)
static thread_local std::vector<int32_t> g_best_counts;

struct Conn { std::vector<int32_t> nb; std::vector<int32_t> w; };      // neighbours in dict insertion order + weights

// The contig stage as an object: candidates arrive in seed order, in one call or in several (the pipelined extension hands
// over the candidates of every rank block as soon as that block is final) -- the state between calls is the state the
// reference's loop carries from one contig to the next (extension_correction.py:358-397).
struct shn_cgraph {
  int k1, r;
  double f;
  FlatMultiMap rmer{1 << 14}, cmer{1 << 14};   // only accepted contigs enter the indexes: start small, grow on demand
  std::vector<Conn> conns;                     // index 0 unused (contigs are 1-based)
  std::vector<int32_t> connw{0};               // scratch counters, one per accepted contig (index 0 unused)
  int32_t idx = 0;
  uint64_t n_evals = 0, n_batches = 0, n_cand_total = 0;
  double t_eval = 0, t_accept = 0;
  shn_cgraph(int k1_, int r_, double f_) : k1(k1_), r(r_), f(f_) { conns.emplace_back(); }

  // accepted[i] = 1-based accepted index of candidate i or 0; bestcnt[i] = hit count of its `best` contig
  void add(const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int32_t* accepted, int32_t* bestcnt) {
    const int C = k1 - 1;
    const bool dbg = getenv("SHN_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (uint64_t i = 0; i < n_cand; i++) { accepted[i] = 0; bestcnt[i] = 0; }
    n_cand_total += n_cand;
    // duplicate_check of one candidate against the current index (read-only on the shared state)
    struct Scratch { std::vector<uint64_t> rk; std::vector<int32_t> hits, dupcnt, touched, cov; };
    auto evaluate = [&](uint64_t c, Scratch& z, int32_t& best_count) -> bool {
      const uint8_t* s = bases + off[c];
      const uint32_t L = (uint32_t)(off[c + 1] - off[c]);
      window_keys(s, L, r, z.rk);
      z.hits.assign(z.rk.size(), -1);
      if (z.dupcnt.size() < (size_t)idx + 1) z.dupcnt.resize((size_t)idx + 1, 0);
      int32_t max_till_now = 0, best = -1;
      for (size_t i = 0; i < z.rk.size(); i++) {
        if (i + 12 < z.rk.size()) rmer.prefetch(z.rk[i + 12]);
        int32_t v = rmer.find(z.rk[i]);
        z.hits[i] = v;
        for (; v != -1; v = rmer.nxt(v)) {
          int32_t d = rmer.va(v);
          if (z.dupcnt[d] == 0) z.touched.push_back(d);
          int32_t cnt = ++z.dupcnt[d];
          if (cnt >= max_till_now) { max_till_now = cnt; best = d; }      // `>=`: the latest wins (:258-259)
        }
      }
      for (int32_t d : z.touched) z.dupcnt[d] = 0;
      z.touched.clear();
      best_count = max_till_now;
      if (best < 0) return false;
      z.cov.assign(L + 1, 0);
      for (size_t i = 0; i < z.rk.size(); i++) {
        bool has = false;
        for (int32_t v = z.hits[i]; v != -1 && !has; v = rmer.nxt(v)) has = rmer.va(v) == best;
        if (has) { z.cov[i] += 1; z.cov[i + r] -= 1; }
      }
      int64_t run = 0, covered = 0;
      for (uint32_t i = 0; i < L; i++) { run += z.cov[i]; if (run > 0) covered++; }
      return (double)covered > f * (double)L;                          // suspect
    };

    std::vector<uint64_t> ck, rk2;
    std::vector<int32_t> newnb;                             // (connw, a member: per accepted contig, shared K-mers with the new contig)
    auto accept = [&](uint64_t c) {
      const uint8_t* s = bases + off[c];
      const uint32_t L = (uint32_t)(off[c + 1] - off[c]);
      idx++;
      accepted[c] = idx;
      conns.emplace_back();
      // contig_connections (:372-397): every earlier contig sharing a K-mer gets +1 per shared position pair, both
      // ways.  The new contig's own dict fills in first-seen order (flat counters, no per-contig hash map); in an
      // earlier contig's dict the new contig can only be the most recent entry.
      window_keys(s, L, C, ck);
      connw.push_back(0);
      newnb.clear();
      for (size_t ki = 0; ki < ck.size(); ki++) {
        const uint64_t key = ck[ki];
        if (ki + 12 < ck.size()) cmer.prefetch(ck[ki + 12]);
        // insert first (one probe): the entries that were there before are the earlier contigs and this contig's own
        // earlier occurrences of the K-mer (skipped)
        int32_t mine = -1;
        for (int32_t v = cmer.add(key, idx, 0, &mine); v != -1 && v != mine; v = cmer.nxt(v)) {
          int32_t c2 = cmer.va(v);
          if (c2 == idx) continue;
          if (connw[c2]++ == 0) newnb.push_back(c2);
          Conn& b = conns[c2];
          if (!b.nb.empty() && b.nb.back() == idx) b.w.back()++;
          else { b.nb.push_back(idx); b.w.push_back(1); }
        }
      }
      Conn& a = conns[idx];
      a.nb = newnb;
      a.w.resize(newnb.size());
      for (size_t j = 0; j < newnb.size(); j++) { a.w[j] = connw[newnb[j]]; connw[newnb[j]] = 0; }
      window_keys(s, L, r, rk2);
      for (size_t i = 0; i < rk2.size(); i++) {
        if (i + 12 < rk2.size()) rmer.prefetch(rk2[i + 12]);
        rmer.add(rk2[i], idx);
      }
    };

    // Candidates are decided in seed order, but a run of candidates with no acceptance among them can be evaluated
    // in parallel against the same index (evaluation is read-only): batches grow while nothing is accepted (duplicates
    // dominate the tail of the seed order) and shrink when something is.  The first accepted candidate of a batch ends
    // it -- the ones after it are evaluated again against the enlarged index.
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_threads = std::min(16u, hw);
    std::vector<Scratch> scratch(n_threads);
    std::vector<uint8_t> susp;
    uint64_t pos = 0, B = 1;
    while (pos < n_cand) {
      const uint64_t end = std::min<uint64_t>(n_cand, pos + B);
      const uint64_t nb = end - pos;
      susp.assign(nb, 0);
      double t0 = dbg ? now() : 0;
      if (nb < 64 || n_threads == 1) {
        for (uint64_t c = pos; c < end; c++) {
          susp[c - pos] = evaluate(c, scratch[0], bestcnt[c]) ? 1 : 0;
          n_evals++;
          if (!susp[c - pos]) { susp.resize(c - pos + 1); break; }         // the rest would be stale anyway
        }
      } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_threads; t++)
          th.emplace_back([&, t]() { for (uint64_t c = pos + t; c < end; c += n_threads) susp[c - pos] = evaluate(c, scratch[t], bestcnt[c]) ? 1 : 0; });
        for (auto& x : th) x.join();
        n_evals += nb;
      }
      n_batches++;
      if (dbg) { double t1 = now(); t_eval += t1 - t0; t0 = t1; }
      uint64_t j = 0;
      while (j < susp.size() && susp[j]) j++;
      if (j < susp.size()) {                        // candidate pos+j is accepted; everything after it is looked at again
        accept(pos + j);
        pos += j + 1;
        B = std::max<uint64_t>(1, B / 2);
      } else {
        pos += susp.size();
        B = std::min<uint64_t>(B * 2, 1024);
      }
      if (dbg) t_accept += now() - t0;
    }
  }
};

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
  static thread_local shn_cgraph* cached = nullptr;      // kept between the sizing call and the fill call
  static thread_local std::vector<int32_t> accepted;
  static thread_local uint64_t cached_n = ~0ULL;
  static thread_local const uint8_t* cached_ptr = nullptr;
  if (!(conn_nb && cached && cached_n == n_cand && cached_ptr == bases)) {
    delete cached;
    cached = new shn_cgraph(k1, r, f);
    accepted.assign(n_cand + 1, 0);
    g_best_counts.assign(n_cand, 0);
    std::vector<int32_t> bc(n_cand + 1, 0);
    cached->add(bases, off, n_cand, accepted.data(), bc.data());
    for (uint64_t i = 0; i < n_cand; i++) g_best_counts[i] = bc[i];
    cached_n = n_cand; cached_ptr = bases;
  }
  int rc = shn_cgraph_sizes(cached, n_acc_out, n_conn);
  if (rc) return rc;
  memcpy(accepted_out, accepted.data(), n_cand * sizeof(int32_t));
  if (conn_nb && conn_w && conn_off) {
    rc = shn_cgraph_export(cached, conn_off, conn_nb, conn_w);
    delete cached;
    cached = nullptr; cached_n = ~0ULL; cached_ptr = nullptr;
    accepted.clear();
  }
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
