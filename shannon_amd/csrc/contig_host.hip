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

// contigs: n_cand candidate strings (bases[off[i]..off[i+1])), in seed order.
// accepted_out[i] = 1-based accepted index or 0.  Connections are returned as CSR in *insertion order*
// (the order Python's dict would iterate): conn_off[n_acc+1], conn_nb[], conn_w[] (caller sizes conn_* with
// the value returned in *n_conn after a first call with conn_nb == NULL).
extern "C" int shn_contig_graph(const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                                int32_t* accepted_out, uint64_t* n_acc_out, uint64_t* conn_off, int32_t* conn_nb, int32_t* conn_w,
                                uint64_t* n_conn) {
  if (!bases || !off || !accepted_out || !n_acc_out || !n_conn) return shn_fail(SHN_ERR_ARG, "shn_contig_graph: NULL argument");
  static thread_local std::vector<Conn> conns;         // kept between the sizing call and the fill call
  static thread_local std::vector<int32_t> accepted;
  std::vector<int32_t>& bestcnt = g_best_counts;
  static thread_local uint64_t cached_n = ~0ULL;
  static thread_local const uint8_t* cached_ptr = nullptr;
  if (!(conn_nb && cached_n == n_cand && cached_ptr == bases)) {
    uint64_t total_bases = off[n_cand] - off[0];
    // only accepted contigs enter the indexes (a few percent of the candidate bases): start small so that the
    // probes of the rejected candidates stay in cache; the maps grow on demand
    (void)total_bases;
    FlatMultiMap rmer(1 << 14), cmer(1 << 14);
    conns.clear(); conns.emplace_back();                 // index 0 unused (contigs are 1-based)
    accepted.assign(n_cand, 0);
    bestcnt.assign(n_cand, 0);
    std::vector<uint64_t> rk, ck;
    std::vector<int32_t> hits;                              // first value index in rmer (or -1) per window
    std::vector<int32_t> dupcnt(1, 0), touched;             // per accepted contig: shared r-mers with the candidate
    std::vector<int32_t> connw(1, 0), newnb;                // per accepted contig: shared K-mers with the new contig
    std::vector<int32_t> cov;
    const int C = k1 - 1;
    int32_t idx = 0;
    const bool dbg = getenv("SHN_DEBUG") != nullptr;
    double tph[4] = {0, 0, 0, 0};
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    uint64_t n_entries = 0;
    for (uint64_t c = 0; c < n_cand; c++) {
      const uint8_t* s = bases + off[c];
      uint32_t L = (uint32_t)(off[c + 1] - off[c]);
      double t0 = dbg ? now() : 0;
      window_keys(s, L, r, rk);
      hits.assign(rk.size(), -1);
      int32_t max_till_now = 0, best = -1;
      for (size_t i = 0; i < rk.size(); i++) {
        if (i + 12 < rk.size()) rmer.prefetch(rk[i + 12]);
        int32_t v = rmer.find(rk[i]);
        hits[i] = v;
        for (; v != -1; v = rmer.nxt(v)) {
          int32_t d = rmer.va(v);
          n_entries++;
          if (dupcnt[d] == 0) touched.push_back(d);
          int32_t cnt = ++dupcnt[d];
          if (cnt >= max_till_now) { max_till_now = cnt; best = d; }      // `>=`: the latest wins (:258-259)
        }
      }
      for (int32_t d : touched) dupcnt[d] = 0;
      touched.clear();
      if (dbg) { double t1 = now(); tph[0] += t1 - t0; t0 = t1; }
      bestcnt[c] = max_till_now;
      bool suspect = false;
      if (best >= 0) {
        cov.assign(L + 1, 0);
        for (size_t i = 0; i < rk.size(); i++) {
          bool has = false;
          for (int32_t v = hits[i]; v != -1 && !has; v = rmer.nxt(v)) has = rmer.va(v) == best;
          if (has) { cov[i] += 1; cov[i + r] -= 1; }
        }
        int64_t run = 0, covered = 0;
        for (uint32_t i = 0; i < L; i++) { run += cov[i]; if (run > 0) covered++; }
        suspect = (double)covered > f * (double)L;
      }
      if (dbg) { double t1 = now(); tph[1] += t1 - t0; t0 = t1; }
      if (suspect) continue;
      idx++;
      accepted[c] = idx;
      dupcnt.push_back(0);
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
      {
        Conn& a = conns[idx];
        a.nb = newnb;
        a.w.resize(newnb.size());
        for (size_t j = 0; j < newnb.size(); j++) { a.w[j] = connw[newnb[j]]; connw[newnb[j]] = 0; }
      }
      if (dbg) { double t1 = now(); tph[2] += t1 - t0; t0 = t1; }
      for (size_t i = 0; i < rk.size(); i++) {
        if (i + 12 < rk.size()) rmer.prefetch(rk[i + 12]);
        rmer.add(rk[i], idx);
      }
      if (dbg) { double t1 = now(); tph[3] += t1 - t0; }
    }
    cached_n = n_cand; cached_ptr = bases;
    if (dbg) fprintf(stderr, "[contig_graph] lookup %.3f s (%llu list entries), coverage %.3f s, connections %.3f s, index insert %.3f s\n", tph[0],
                     (unsigned long long)n_entries, tph[1], tph[2], tph[3]);
  }
  uint64_t n_acc = conns.size() - 1, total = 0;
  for (uint64_t i = 1; i <= n_acc; i++) total += conns[i].nb.size();
  *n_acc_out = n_acc;
  *n_conn = total;
  memcpy(accepted_out, accepted.data(), n_cand * sizeof(int32_t));
  if (conn_nb && conn_w && conn_off) {
    uint64_t p = 0;
    for (uint64_t i = 1; i <= n_acc; i++) {
      conn_off[i - 1] = p;
      for (size_t j = 0; j < conns[i].nb.size(); j++) { conn_nb[p] = conns[i].nb[j]; conn_w[p] = conns[i].w[j]; p++; }
    }
    conn_off[n_acc] = p;
    cached_n = ~0ULL; cached_ptr = nullptr;
    conns.clear(); accepted.clear();
  }
  return SHN_OK;
}

// hit count of the `best` contig (max_till_now, extension_correction.py:255-259) of every candidate of this thread's
// last shn_contig_graph call: what the component-sharded duplicate check needs to rule out interference from other shards
extern "C" int shn_contig_best_counts(int32_t* out, uint64_t n_cand) {
  if (!out) return shn_fail(SHN_ERR_ARG, "shn_contig_best_counts: NULL argument");
  if (g_best_counts.size() != n_cand) return shn_fail(SHN_ERR_ARG, "shn_contig_best_counts: no matching shn_contig_graph call");
  memcpy(out, g_best_counts.data(), n_cand * sizeof(int32_t));
  return SHN_OK;
}
