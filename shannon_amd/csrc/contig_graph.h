// Shared by contig_host.hip (the sequential contig stage) and contig_gpu.hip (the GPU-clustered one): duplicate_check
// (extension_correction.py:247-270) and the contig graph by shared K-mers (:372-397) over candidates in seed order.
#pragma once
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
static inline void window_keys(const uint8_t* s, uint32_t L, int k, std::vector<uint64_t>& out) {
  out.clear();
  if ((int)L < k) return;
  uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1), v = 0;
  for (uint32_t i = 0; i < L; i++) {
    v = ((v << 2) | (uint64_t)(code_of(s[i]) & 3)) & mask;
    if ((int)i >= k - 1) out.push_back(v);
  }
}

// per candidate of the last shn_contig_graph call of this thread: the hit count of its `best` contig (0: no hit)
extern thread_local std::vector<int32_t> g_best_counts;

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
    const unsigned hw = (unsigned)shn_host_cpus();
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

