// Native host implementation of the final containment de-duplication, faster_reps.find_reps
// (faster_reps.py:98-131 with duplicate_check_ends :60-92), row a31: every 24-mer of every transcript is
// indexed, a transcript is dropped when its first and last 24-mer lie on another transcript at the right
// distance (|diff - (len-24)| < 3) and that one is longer (or equal and name-smaller).  Host code: the
// record set is the assembler's output, not the read set; SURVEY.md 8f ranks a GPU version as "next".
#include "common.h"
#include "flatmap.h"
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

static inline int pcode(uint8_t c) {
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// names/seqs: n records (ASCII, offsets n+1 each), in file order.  keep_out[i] = 1 if record i survives.
// A name that occurs more than once behaves like the reference's dict (the later sequence replaces the
// earlier one, keep_out of the earlier record is 0).  Returns SHN_ERR_ARG if a sequence holds a non-ACGT base.
extern "C" int shn_find_reps(const uint8_t* names, const uint64_t* name_off, const uint8_t* seqs, const uint64_t* seq_off,
                             uint64_t n, int ds, int r, uint8_t* keep_out) {
  if ((n && (!names || !name_off || !seqs || !seq_off)) || !keep_out) return shn_fail(SHN_ERR_ARG, "shn_find_reps: NULL argument");
  if (r < 1 || r > 32) return shn_fail(SHN_ERR_ARG, "shn_find_reps: r must be in [1,32]");
  StringInterner ids(n + 16);
  std::vector<int64_t> rec_of;                  // name id -> latest record
  std::vector<int32_t> id_of(n);
  const uint64_t mask = r == 32 ? ~0ULL : ((1ULL << (2 * r)) - 1);
  for (uint64_t i = 0; i < n; i++) {
    bool is_new;
    int32_t id = ids.intern((const char*)names + name_off[i], name_off[i + 1] - name_off[i], &is_new);
    if (is_new) rec_of.push_back((int64_t)i); else rec_of[id] = (int64_t)i;
    id_of[i] = id;
  }
  auto key_of = [&](const uint8_t* s, uint64_t L, uint64_t pos, bool rc, uint64_t& key) {
    // r-mer at `pos` of the sequence (rc: of its reverse complement)
    key = 0;
    for (int j = 0; j < r; j++) {
      int c = rc ? 3 - pcode(s[L - 1 - (pos + j)]) : pcode(s[pos + j]);
      key = (key << 2) | (uint64_t)(c & 3);
    }
  };
  // The reference indexes every r-mer of every transcript (faster_reps.py:104-112) but only ever looks up the first and the
  // last r-mer of each record, plain and reverse-complemented.  So: those (at most 4 per record) form a small query set, all
  // records are scanned on host threads for occurrences of query keys (a rolling key, a probe of a cache-resident set), and
  // only these occurrences enter the index -- in record order, as the reference appends them.
  {
    std::atomic<int> bad{0};
    const unsigned nt0 = std::max(1u, std::min(32u, (unsigned)shn_host_cpus()));
    const unsigned nt = n < 4096 ? 1 : nt0;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t]() {
      for (uint64_t i = n * t / nt; i < n * (t + 1) / nt; i++)
        for (uint64_t p = seq_off[i]; p < seq_off[i + 1]; p++) if (pcode(seqs[p]) < 0) { bad.store(1); return; }
    });
    for (auto& x : th) x.join();
    if (bad.load()) return shn_fail(SHN_ERR_ARG, "shn_find_reps: non-ACGT base in a transcript");
  }
  size_t qcap = 1024;
  while (qcap < rec_of.size() * 16) qcap <<= 1;
  std::vector<uint64_t> qset(qcap, ~0ULL);        // open addressing; ~0 = empty (an r-mer of r < 32 never is; r = 32: all-T handled below)
  bool q_all_t = false;
  auto q_add = [&](uint64_t key) {
    if (key == ~0ULL) { q_all_t = true; return; }
    size_t sl = fm_mix(key) & (qcap - 1);
    while (qset[sl] != ~0ULL && qset[sl] != key) sl = (sl + 1) & (qcap - 1);
    qset[sl] = key;
  };
  auto q_has = [&](uint64_t key) -> bool {
    if (key == ~0ULL) return q_all_t;
    size_t sl = fm_mix(key) & (qcap - 1);
    while (qset[sl] != ~0ULL) { if (qset[sl] == key) return true; sl = (sl + 1) & (qcap - 1); }
    return false;
  };
  for (size_t id = 0; id < rec_of.size(); id++) {
    const uint64_t i = (uint64_t)rec_of[id];
    const uint8_t* s = seqs + seq_off[i];
    const uint64_t L = seq_off[i + 1] - seq_off[i];
    if (L < (uint64_t)r) continue;
    for (int flip = 0; flip < (ds ? 2 : 1); flip++) {
      uint64_t kf, kl;
      key_of(s, L, 0, flip == 1, kf);
      key_of(s, L, L - r, flip == 1, kl);
      q_add(kf); q_add(kl);
    }
  }
  struct Occ { uint64_t key; int32_t id, pos; };
  const unsigned nthr = n < 4096 ? 1 : std::max(1u, std::min(32u, (unsigned)shn_host_cpus()));
  std::vector<std::vector<Occ>> found(nthr);
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; t++) th.emplace_back([&, t]() {
      std::vector<Occ>& out = found[t];
      for (uint64_t i = n * t / nthr; i < n * (t + 1) / nthr; i++) {
        const uint8_t* s = seqs + seq_off[i];
        const uint64_t L = seq_off[i + 1] - seq_off[i];
        uint64_t key = 0;
        for (uint64_t p = 0; p < L; p++) {
          key = ((key << 2) | (uint64_t)pcode(s[p])) & mask;
          if (p + 1 >= (uint64_t)r && q_has(key)) out.push_back(Occ{key, id_of[i], (int32_t)(p + 1 - r)});
        }
      }
    });
    for (auto& x : th) x.join();
  }
  size_t n_occ = 0;
  for (auto& v : found) n_occ += v.size();
  FlatMultiMap index(n_occ + 1024);
  for (auto& v : found) for (const Occ& o : v) index.add(o.key, o.id, o.pos);
  memset(keep_out, 0, n);
  std::vector<std::pair<int32_t, std::pair<int64_t, int64_t>>> pos;    // (other name id, (first pos, last pos)) in first-seen order
  for (size_t id = 0; id < rec_of.size(); id++) {
    uint64_t i = (uint64_t)rec_of[id];
    const uint8_t* s = seqs + seq_off[i];
    uint64_t L = seq_off[i + 1] - seq_off[i];
    bool drop = false;
    for (int flip = 0; flip < (ds ? 2 : 1) && !drop; flip++) {
      if (L < (uint64_t)r) continue;
      uint64_t kf, kl;
      key_of(s, L, 0, flip == 1, kf);
      key_of(s, L, L - r, flip == 1, kl);
      int32_t vf = index.find(kf), vl = index.find(kl);
      if (vf == -1 || vl == -1) continue;
      pos.clear();
      auto slot = [&](int32_t o) -> std::pair<int64_t, int64_t>& {
        for (auto& e : pos) if (e.first == o) return e.second;
        pos.push_back({o, {-2, -2}});                      // -2 = "key not created by this list"
        return pos.back().second;
      };
      for (int32_t v = vf; v != -1; v = index.nxt(v)) {
        int32_t o = index.va(v);
        if (o == (int32_t)id) continue;
        auto& e = slot(o);
        if (e.first == -2 && e.second == -2) e = {index.vb(v), -1}; else e.first = index.vb(v);
      }
      for (int32_t v = vl; v != -1; v = index.nxt(v)) {
        int32_t o = index.va(v);
        if (o == (int32_t)id) continue;
        auto& e = slot(o);
        if (e.first == -2 && e.second == -2) e = {-1, index.vb(v)}; else e.second = index.vb(v);
      }
      for (auto& e : pos) {
        int64_t p0 = e.second.first, p1 = e.second.second;
        if (p0 < 0 || p1 < 0) continue;
        int64_t diff = p1 - p0;
        int64_t d = diff - ((int64_t)L - r);
        if (d < 0) d = -d;
        if (d >= 3) continue;
        uint64_t oi = (uint64_t)rec_of[e.first];
        uint64_t OL = seq_off[oi + 1] - seq_off[oi];
        bool name_gt = false;
        if (L == OL) {
          size_t la = ids.len((int32_t)id), lb = ids.len(e.first);
          int c = memcmp(ids.data((int32_t)id), ids.data(e.first), std::min(la, lb));
          name_gt = c > 0 || (c == 0 && la > lb);
        }
        if (L < OL || name_gt) { drop = true; break; }
      }
    }
    if (!drop) keep_out[i] = 1;
  }
  return SHN_OK;
}
