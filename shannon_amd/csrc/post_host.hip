// Native host implementation of the final containment de-duplication, faster_reps.find_reps
// (faster_reps.py:98-131 with duplicate_check_ends :60-92), row a31: every 24-mer of every transcript is
// indexed, a transcript is dropped when its first and last 24-mer lie on another transcript at the right
// distance (|diff - (len-24)| < 3) and that one is longer (or equal and name-smaller).  Host code: the
// record set is the assembler's output, not the read set; SURVEY.md 8f ranks a GPU version as "next".
#include "common.h"
#include "flatmap.h"
#include "post_dev.h"
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

struct PCodeLut { int8_t v[256]; PCodeLut() { memset(v, -1, 256); v['A'] = 0; v['C'] = 1; v['G'] = 2; v['T'] = 3; } };
static const PCodeLut kPCode;
static inline int pcode(uint8_t c) { return kPCode.v[c]; }                // (a table: a switch mispredicts on every other base)

// names/seqs: n records (ASCII, offsets n+1 each), in file order.  keep_out[i] = 1 if record i survives.
// A name that occurs more than once behaves like the reference's dict (the later sequence replaces the
// earlier one, keep_out of the earlier record is 0).  Returns SHN_ERR_ARG if a sequence holds a non-ACGT base.
static int find_reps_core(const uint8_t* const* nptr, const uint64_t* nlen, const uint8_t* const* sptr, const uint64_t* slen, uint64_t n, int ds, int r,
                          uint8_t* keep_out, PostDev* dev = nullptr, const uint64_t* goff = nullptr);
extern "C" int shn_find_reps(const uint8_t* names, const uint64_t* name_off, const uint8_t* seqs, const uint64_t* seq_off,
                             uint64_t n, int ds, int r, uint8_t* keep_out) {
  if ((n && (!names || !name_off || !seqs || !seq_off)) || !keep_out) return shn_fail(SHN_ERR_ARG, "shn_find_reps: NULL argument");
  std::vector<const uint8_t*> np_(n), sp_(n);
  std::vector<uint64_t> nl_(n), sl_(n);
  for (uint64_t i = 0; i < n; i++) { np_[i] = names + name_off[i]; nl_[i] = name_off[i + 1] - name_off[i]; sp_[i] = seqs + seq_off[i]; sl_[i] = seq_off[i + 1] - seq_off[i]; }
  return find_reps_core(np_.data(), nl_.data(), sp_.data(), sl_.data(), n, ds, r, keep_out);
}
// (records given by pointer + length: shn_post_finalize hands over lines of the text it was given, without packing them)
// dev + goff (global offset of every record's sequence in the uploaded text): the scan for the query r-mers runs on the device
static int find_reps_core(const uint8_t* const* nptr, const uint64_t* nlen, const uint8_t* const* sptr, const uint64_t* slen, uint64_t n, int ds, int r,
                          uint8_t* keep_out, PostDev* dev, const uint64_t* goff) {
  if (r < 1 || r > 32) return shn_fail(SHN_ERR_ARG, "shn_find_reps: r must be in [1,32]");
  const bool dbgf = getenv("SHN_DEBUG") != nullptr || getenv("SHN_POST_LAPS") != nullptr;
  auto nowf = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
  const double tf0 = nowf();
  StringInterner ids(n + 16);
  std::vector<int64_t> rec_of;                  // name id -> latest record
  std::vector<int32_t> id_of(n);
  const uint64_t mask = r == 32 ? ~0ULL : ((1ULL << (2 * r)) - 1);
  for (uint64_t i = 0; i < n; i++) {
    bool is_new;
    int32_t id = ids.intern((const char*)nptr[i], nlen[i], &is_new);
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
  // (bases outside ACGT are looked for by the scan below, which reads every base anyway)
  std::atomic<int> bad{0};
  const double tf1 = nowf();
  size_t qcap = 1024;
  while (qcap < rec_of.size() * 16) qcap <<= 1;
  std::vector<uint64_t> qset(qcap, ~0ULL);        // open addressing; ~0 = empty (an r-mer of r < 32 never is; r = 32: all-T handled below)
  bool q_all_t = false;
  // in front of the set a 1 MB bit table that stays in a core's cache: 19 of 20 window positions end there
  std::vector<uint64_t> qbits(1u << 17, 0);
  std::vector<uint64_t> qlist;                    // (device scan: the query keys as a list)
  auto q_add = [&](uint64_t key) {
    if (dev) qlist.push_back(key);
    if (key == ~0ULL) { q_all_t = true; return; }
    { const uint64_t h = fm_mix(key) >> 41; qbits[h >> 6] |= 1ULL << (h & 63); }
    size_t sl = fm_mix(key) & (qcap - 1);
    while (qset[sl] != ~0ULL && qset[sl] != key) sl = (sl + 1) & (qcap - 1);
    qset[sl] = key;
  };
  auto q_has = [&](uint64_t key) -> bool {
    if (key == ~0ULL) return q_all_t;
    const uint64_t hm = fm_mix(key);
    { const uint64_t h = hm >> 41; if (!((qbits[h >> 6] >> (h & 63)) & 1)) return false; }
    size_t sl = hm & (qcap - 1);
    while (qset[sl] != ~0ULL) { if (qset[sl] == key) return true; sl = (sl + 1) & (qcap - 1); }
    return false;
  };
  // the first / last r-mer of every record, plain and reverse-complemented: computed on the host threads (and kept for the decisions
  // below), entered into the set in record order
  const size_t n_ids = rec_of.size();
  const int n_flip = ds ? 2 : 1;
  std::vector<uint64_t> qkeys(n_ids * 4, 0);                      // [id][flip][first, last]
  const unsigned nthr_q = n_ids < 8192 ? 1 : std::max(1u, std::min(16u, (unsigned)shn_host_cpus()));
  {
    auto work = [&](size_t lo, size_t hi) {
      for (size_t id = lo; id < hi; id++) {
        const uint64_t i = (uint64_t)rec_of[id];
        const uint8_t* s = sptr[i];
        const uint64_t L = slen[i];
        if (L < (uint64_t)r) continue;
        for (int flip = 0; flip < n_flip; flip++) { key_of(s, L, 0, flip == 1, qkeys[id * 4 + flip * 2]); key_of(s, L, L - r, flip == 1, qkeys[id * 4 + flip * 2 + 1]); }
      }
    };
    if (nthr_q <= 1) work(0, n_ids);
    else { std::vector<std::thread> th; for (unsigned t = 0; t < nthr_q; t++) th.emplace_back(work, n_ids * t / nthr_q, n_ids * (t + 1) / nthr_q); for (auto& x : th) x.join(); }
  }
  for (size_t id = 0; id < n_ids; id++) {
    if (slen[(uint64_t)rec_of[id]] < (uint64_t)r) continue;
    for (int flip = 0; flip < n_flip; flip++) { q_add(qkeys[id * 4 + flip * 2]); q_add(qkeys[id * 4 + flip * 2 + 1]); }
  }
  const double tf2 = nowf();
  struct Occ { uint64_t key; int32_t id, pos; };
  const unsigned nthr = n < 4096 ? 1 : std::max(1u, std::min(32u, (unsigned)shn_host_cpus()));
  std::vector<std::vector<Occ>> found(nthr);
  bool scanned = false;
  if (dev && goff && r < 32 && n) {
    // the occurrences from the device, in record order and by position inside a record -- the order the host scan appends them in
    std::vector<uint32_t> lens(n);
    bool fits = true;
    for (uint64_t i = 0; i < n; i++) { if (slen[i] > 0xFFFFFFF0ULL) fits = false; lens[i] = (uint32_t)slen[i]; }
    if (fits) {
      std::vector<uint32_t> hr, hp;
      std::vector<uint64_t> hk;
      int dbad = 0;
      uint64_t total = 0;
      for (uint64_t i = 0; i < n; i++) total += slen[i];
      const int rcs = post_dev_scan(dev, goff, lens.data(), n, r, qlist.data(), qlist.size(), std::min<uint64_t>(total + 1, 1ULL << 24), hr, hp, hk, &dbad);
      if (rcs == SHN_OK) {
        if (dbad) return shn_fail(SHN_ERR_ARG, "shn_find_reps: non-ACGT base in a transcript");
        found[0].reserve(hr.size());
        for (size_t h = 0; h < hr.size(); h++) found[0].push_back(Occ{hk[h], id_of[hr[h]], (int32_t)hp[h]});
        scanned = true;
      } else if (rcs != SHN_ERR_OVERFLOW) return rcs;          // (more hits than the buffer: the host scan takes over)
    }
  }
  if (!scanned) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; t++) th.emplace_back([&, t]() {
      std::vector<Occ>& out = found[t];
      for (uint64_t i = n * t / nthr; i < n * (t + 1) / nthr; i++) {
        const uint8_t* s = sptr[i];
        const uint64_t L = slen[i];
        uint64_t key = 0;
        int any_bad = 0;
        for (uint64_t p = 0; p < L; p++) {
          const int c = pcode(s[p]);
          any_bad |= c;                                   // (-1 sets the sign bit)
          key = ((key << 2) | (uint64_t)(c & 3)) & mask;
          if (p + 1 >= (uint64_t)r && q_has(key)) out.push_back(Occ{key, id_of[i], (int32_t)(p + 1 - r)});
        }
        if (any_bad < 0) { bad.store(1); return; }
      }
    });
    for (auto& x : th) x.join();
  }
  if (bad.load()) return shn_fail(SHN_ERR_ARG, "shn_find_reps: non-ACGT base in a transcript");
  const double tf3 = nowf();
  size_t n_occ = 0;
  for (auto& v : found) n_occ += v.size();
  FlatMultiMap index(n_occ + 1024);
  for (auto& v : found) for (const Occ& o : v) index.add(o.key, o.id, o.pos);
  memset(keep_out, 0, n);
  const double tf4 = nowf();
  // (a record's decision reads the index, the names and the lengths -- nothing another decision writes: the records on the host threads)
  auto decide_range = [&](size_t id_lo, size_t id_hi) {
  std::vector<std::pair<int32_t, std::pair<int64_t, int64_t>>> pos;    // (other name id, (first pos, last pos)) in first-seen order
  for (size_t id = id_lo; id < id_hi; id++) {
    uint64_t i = (uint64_t)rec_of[id];
    uint64_t L = slen[i];
    bool drop = false;
    for (int flip = 0; flip < (ds ? 2 : 1) && !drop; flip++) {
      if (L < (uint64_t)r) continue;
      const uint64_t kf = qkeys[id * 4 + flip * 2], kl = qkeys[id * 4 + flip * 2 + 1];
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
        uint64_t OL = slen[oi];
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
  };
  if (nthr_q <= 1) decide_range(0, n_ids);
  else { std::vector<std::thread> th; for (unsigned t = 0; t < nthr_q; t++) th.emplace_back(decide_range, n_ids * t / nthr_q, n_ids * (t + 1) / nthr_q); for (auto& x : th) x.join(); }
  if (dbgf) fprintf(stderr, "[find_reps] names+check %.3f, query set %.3f, scan %.3f, index %.3f (%zu occurrences), decide %.3f s\n", tf1 - tf0, tf2 - tf1,
                    tf3 - tf2, tf4 - tf3, n_occ, nowf() - tf4);
  return SHN_OK;
}

// ---- the whole merge in one call: process_concatenated_fasta.py:6-32, the length sort of shannon.py:603 and find_reps above,
// over the text of all_reconstructed.fasta (every partition's reconstructed.fasta + reconstructed_single_contigs.fasta).  At
// BASELINE configs[2] that file has 120 k records / 150 MB; the line-by-line Python form of these three steps was 0.6 s of
// the step.  Semantics kept, quirks included: a repeated name N becomes "N_<count>" glued to its remaining header fields
// (no separator), sequences of at most 199 bases (line length <= 200 with its newline) are dropped, a sequence (or, double
// stranded, its reverse complement) seen before is dropped, a header line seen twice keeps its last sequence, records are
// ordered by (sequence line length, header line).
#include <map>
#include <mutex>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <string>

struct shn_post {
  std::string names, seqs;
  std::vector<uint64_t> name_off{0}, seq_off{0};
};

extern "C" void shn_post_destroy(shn_post* p) { delete p; }
extern "C" uint64_t shn_post_count(const shn_post* p) { return p ? p->name_off.size() - 1 : 0; }
extern "C" int shn_post_sizes(const shn_post* p, uint64_t* name_bytes, uint64_t* seq_bytes) {
  if (!p || !name_bytes || !seq_bytes) return shn_fail(SHN_ERR_ARG, "shn_post_sizes: NULL argument");
  *name_bytes = p->names.size(); *seq_bytes = p->seqs.size();
  return SHN_OK;
}
extern "C" int shn_post_export(const shn_post* p, uint8_t* names, uint64_t* name_off, uint8_t* seqs, uint64_t* seq_off) {
  if (!p || !names || !name_off || !seqs || !seq_off) return shn_fail(SHN_ERR_ARG, "shn_post_export: NULL argument");
  memcpy(names, p->names.data(), p->names.size());
  memcpy(seqs, p->seqs.data(), p->seqs.size());
  memcpy(name_off, p->name_off.data(), p->name_off.size() * 8);
  memcpy(seq_off, p->seq_off.data(), p->seq_off.size() * 8);
  return SHN_OK;
}

static int post_finalize_impl(shn_ctx* ctx, const uint8_t* const* bufs, const uint64_t* lens, uint64_t n_bufs, int ds, int r, shn_post** out);
extern "C" int shn_post_finalize_bufs(const uint8_t* const* bufs, const uint64_t* lens, uint64_t n_bufs, int ds, int r, shn_post** out) {
  return post_finalize_impl(nullptr, bufs, lens, n_bufs, ds, r, out);
}
// the same with the two passes over the bases on the device of `ctx` (csrc/post_gpu.hip): fingerprints of every sequence line and
// of its reverse complement, and the scan of the surviving records for the query r-mers of find_reps
extern "C" int shn_post_finalize_dev(shn_ctx* ctx, const uint8_t* const* bufs, const uint64_t* lens, uint64_t n_bufs, int ds, int r, shn_post** out) {
  if (!ctx) return shn_fail(SHN_ERR_ARG, "shn_post_finalize_dev: ctx is NULL");
  return post_finalize_impl(ctx, bufs, lens, n_bufs, ds, r, out);
}
extern "C" int shn_post_finalize(const uint8_t* text, uint64_t n_bytes, int ds, int r, shn_post** out) {
  if ((n_bytes && !text) || !out) return shn_fail(SHN_ERR_ARG, "shn_post_finalize: NULL argument");
  return shn_post_finalize_bufs(&text, &n_bytes, 1, ds, r, out);
}
// ---- the merge in pieces.  A piece = one text of all_reconstructed.fasta (the single contigs, one partition's transcripts) whose last
// line ends: its lines are found, its bytes go to the device, its sequence lines are fingerprinted -- independent of every other piece,
// so a partition's piece is prepared by the host thread that has just made it, beside the graph stage of the partitions still
// running (shn_post_stream_add); the order-dependent rules run over the pieces in index order at the end (shn_post_stream_finish).
typedef std::string_view SV;
static inline bool post_is_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
static inline SV post_strip(SV s) { size_t a = 0, b = s.size(); while (a < b && post_is_ws((uint8_t)s[a])) a++; while (b > a && post_is_ws((uint8_t)s[b - 1])) b--; return s.substr(a, b - a); }
struct PostPiece {
  const uint8_t* text = nullptr; uint64_t n = 0, dev_base = 0;
  std::vector<SV> lines;
  std::vector<uint64_t> line_goff;        // offset of every line in the device text
  std::vector<uint64_t> fp;               // 4 words per line (0 for lines that are not sequence lines); empty: no device
  std::vector<uint8_t> acgt;              // per line: 1 = a sequence line of A C G T only (known from the fingerprint pass)
};
// dev == NULL: lines only.  on: the context whose stream carries the upload and the fingerprint kernel.
static int post_prep_piece(PostDev* dev, shn_ctx* on, PostPiece& P) {
  const uint8_t* text = P.text;
  for (uint64_t p = 0; p < P.n;) {
    const void* q = memchr(text + p, '\n', P.n - p);
    const uint64_t e = q ? (uint64_t)((const uint8_t*)q - text) + 1 : P.n;
    P.lines.push_back(SV((const char*)text + p, e - p));
    P.line_goff.push_back(P.dev_base + p);
    p = e;
  }
  if (!dev || P.lines.empty()) return SHN_OK;
  int rcd = post_dev_upload(dev, on, P.dev_base, P.text, P.n);
  if (rcd) return rcd;
  std::vector<uint64_t> so; std::vector<uint32_t> sl; std::vector<uint32_t> which;
  for (size_t i = 0; i < P.lines.size(); i++) {
    const SV line = P.lines[i];
    size_t a = 0;
    while (a < line.size() && post_is_ws((uint8_t)line[a])) a++;
    if (a == line.size() || line[a] == '>' || line.size() <= 200) continue;
    const SV cur = post_strip(line);
    so.push_back(P.line_goff[i] + (uint64_t)(cur.data() - line.data())); sl.push_back((uint32_t)cur.size()); which.push_back((uint32_t)i);
  }
  std::vector<uint64_t> got(so.size() * 4);
  std::vector<uint8_t> other(so.size() + 1);
  rcd = post_dev_fingerprints(dev, so.data(), sl.data(), so.size(), got.data(), on, other.data());
  if (rcd) return rcd;
  P.fp.assign(P.lines.size() * 4, 0);
  P.acgt.assign(P.lines.size(), 0);
  for (size_t j = 0; j < which.size(); j++) { memcpy(&P.fp[(size_t)which[j] * 4], &got[j * 4], 32); P.acgt[which[j]] = other[j] ? 0 : 1; }
  if (so.empty()) { HIP_TRY(hipStreamSynchronize(on->stream)); }          // (the fingerprint call waits for the stream; without one the upload is waited for here)
  return SHN_OK;
}
static int post_finalize_lines(std::vector<SV>& lines, std::vector<uint64_t>& line_goff, std::vector<uint64_t>& fp, bool use_fp, PostDev* dev, int ds, int r,
                               shn_post** out, const std::vector<uint8_t>* acgt = nullptr);

struct shn_post_stream {
  shn_ctx* ctx = nullptr;
  PostDev* dev = nullptr;
  std::mutex mu;
  uint64_t cursor = 0;
  std::map<uint64_t, PostPiece*> pieces;
  ~shn_post_stream() { for (auto& kv : pieces) delete kv.second; if (dev) post_dev_destroy(dev); }
};
extern "C" void shn_post_stream_destroy(shn_post_stream* ps) { delete ps; }
extern "C" int shn_post_stream_begin(shn_ctx* ctx, uint64_t capacity, shn_post_stream** out) {
  if (!ctx || !out) return shn_fail(SHN_ERR_ARG, "shn_post_stream_begin: NULL argument");
  *out = nullptr;
  shn_post_stream* ps = new shn_post_stream();
  ps->ctx = ctx;
  int rc = post_dev_create_cap(ctx, std::max<uint64_t>(capacity, 1u << 20), &ps->dev);
  if (rc) { delete ps; return rc; }
  *out = ps;
  return SHN_OK;
}
// piece `index` of the concatenation (any order, any host thread, every index once); the text must stay where it is until the
// stream is finished or destroyed.  SHN_ERR_OVERFLOW: no room left in the device text (the caller merges in one piece instead).
extern "C" int shn_post_stream_add(shn_post_stream* ps, uint64_t index, const uint8_t* text, uint64_t n_bytes) {
  if (!ps || (n_bytes && !text)) return shn_fail(SHN_ERR_ARG, "shn_post_stream_add: NULL argument");
  if (n_bytes && text[n_bytes - 1] != '\n') return shn_fail(SHN_ERR_ARG, "shn_post_stream_add: a piece must end its last line");
  PostPiece* P = new PostPiece();
  P->text = text; P->n = n_bytes;
  {
    std::lock_guard<std::mutex> lk(ps->mu);
    if (ps->pieces.count(index)) { delete P; return shn_fail(SHN_ERR_ARG, "shn_post_stream_add: piece given twice"); }
    if (ps->cursor + n_bytes > post_dev_capacity(ps->dev)) { delete P; return shn_fail(SHN_ERR_OVERFLOW, "shn_post_stream_add: the device text is full"); }
    P->dev_base = ps->cursor;
    ps->cursor += n_bytes;
    ps->pieces[index] = P;
  }
  return post_prep_piece(ps->dev, shn_thread_ctx(ps->ctx), *P);
}
extern "C" int shn_post_stream_finish(shn_post_stream* ps, int ds, int r, shn_post** out) {
  if (!ps || !out) return shn_fail(SHN_ERR_ARG, "shn_post_stream_finish: NULL argument");
  std::vector<SV> lines;
  std::vector<uint64_t> line_goff, fp;
  std::vector<uint8_t> acgt;
  size_t nl = 0;
  for (auto& kv : ps->pieces) nl += kv.second->lines.size();
  lines.reserve(nl); line_goff.reserve(nl); fp.reserve(nl * 4);
  for (auto& kv : ps->pieces) {                                          // (a std::map: in index order)
    PostPiece& P = *kv.second;
    lines.insert(lines.end(), P.lines.begin(), P.lines.end());
    line_goff.insert(line_goff.end(), P.line_goff.begin(), P.line_goff.end());
    if (P.fp.size() == P.lines.size() * 4) fp.insert(fp.end(), P.fp.begin(), P.fp.end()); else fp.insert(fp.end(), P.lines.size() * 4, 0);
    if (P.acgt.size() == P.lines.size()) acgt.insert(acgt.end(), P.acgt.begin(), P.acgt.end()); else acgt.insert(acgt.end(), P.lines.size(), 0);
  }
  return post_finalize_lines(lines, line_goff, fp, true, ps->dev, ds, r, out, &acgt);
}

// the same over the concatenation of several buffers (the per-partition FASTA texts as they come out of shn_sparse_flow)
static int post_finalize_impl(shn_ctx* ctx, const uint8_t* const* bufs, const uint64_t* lens, uint64_t n_bufs, int ds, int r, shn_post** out) {
  if (!out || (n_bufs && (!bufs || !lens))) return shn_fail(SHN_ERR_ARG, "shn_post_finalize: NULL argument");
  for (uint64_t i = 0; i < n_bufs; i++) if (lens[i] && !bufs[i]) return shn_fail(SHN_ERR_ARG, "shn_post_finalize: NULL buffer");
  // a buffer that does not end its last line would run into the next one: join them all in that (unusual) case
  std::string joined;
  std::vector<std::pair<const uint8_t*, uint64_t>> pieces;
  bool clean = true;
  for (uint64_t i = 0; i + 1 < n_bufs; i++) if (lens[i] && bufs[i][lens[i] - 1] != '\n') clean = false;
  if (clean) { for (uint64_t i = 0; i < n_bufs; i++) if (lens[i]) pieces.push_back({bufs[i], lens[i]}); }
  else {
    for (uint64_t i = 0; i < n_bufs; i++) joined.append((const char*)bufs[i], lens[i]);
    pieces.push_back({(const uint8_t*)joined.data(), joined.size()});
  }
  uint64_t total = 0;
  for (auto& pc : pieces) total += pc.second;
  PostDev* dev = nullptr;
  struct DevFree { PostDev*& d; ~DevFree() { if (d) post_dev_destroy(d); } } dev_free{dev};
  if (ctx && total) { int rcd = post_dev_create_cap(ctx, total, &dev); if (rcd) return rcd; }
  std::vector<SV> lines;
  std::vector<uint64_t> line_goff, fp;
  std::vector<uint8_t> acgt;
  uint64_t base = 0;
  for (auto& pc : pieces) {
    PostPiece P;
    P.text = pc.first; P.n = pc.second; P.dev_base = base;
    base += pc.second;
    int rcd = post_prep_piece(dev, ctx, P);
    if (rcd) return rcd;
    lines.insert(lines.end(), P.lines.begin(), P.lines.end());
    line_goff.insert(line_goff.end(), P.line_goff.begin(), P.line_goff.end());
    if (dev) {
      if (P.fp.size() == P.lines.size() * 4) fp.insert(fp.end(), P.fp.begin(), P.fp.end()); else fp.insert(fp.end(), P.lines.size() * 4, 0);
      if (P.acgt.size() == P.lines.size()) acgt.insert(acgt.end(), P.acgt.begin(), P.acgt.end()); else acgt.insert(acgt.end(), P.lines.size(), 0);
    }
  }
  return post_finalize_lines(lines, line_goff, fp, dev != nullptr, dev, ds, r, out, dev ? &acgt : nullptr);
}
static int post_finalize_lines(std::vector<SV>& lines, std::vector<uint64_t>& line_goff, std::vector<uint64_t>& fp, bool use_fp, PostDev* dev, int ds, int r,
                               shn_post** out, const std::vector<uint8_t>* acgt) {
  auto is_ws = [](uint8_t c) { return post_is_ws(c); };
  auto strip = [&](SV s) { return post_strip(s); };
  // ---- process_concatenated
  std::vector<std::string> own;                        // renamed header lines (stable addresses: reserved below)
  std::vector<std::pair<SV, SV>> recs;                 // (header line, sequence line), lines with their newline
  std::vector<uint32_t> rec_li;                        // ... and the index of the sequence line
  const double tq0 = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }();
  std::unordered_map<SV, uint64_t> seen;
  uint8_t comp[256];
  for (int c = 0; c < 256; c++) comp[c] = (uint8_t)c;
  comp['A'] = 'T'; comp['C'] = 'G'; comp['G'] = 'C'; comp['T'] = 'A';
  auto revcomp_into = [&](SV cur, std::string& buf) {
    buf.resize(cur.size());
    const char* src = cur.data() + cur.size() - 1; char* dst = &buf[0];
    for (size_t i = 0, m = cur.size(); i < m; i++) dst[i] = (char)comp[(uint8_t)src[-(ptrdiff_t)i]];
  };
  own.reserve(lines.size() + 2);                      // (renamed header lines: at most one per line; the reservation keeps their addresses)
  std::vector<uint64_t> hf(lines.size(), 0), hr(lines.size(), 0);
  if (use_fp) { for (size_t i = 0; i < lines.size(); i++) { hf[i] = fp[4 * i] ^ (fp[4 * i + 1] * 0x9E3779B97F4A7C15ULL); hr[i] = fp[4 * i + 2] ^ (fp[4 * i + 3] * 0x9E3779B97F4A7C15ULL); } }
  else {
    const unsigned nt = lines.size() < 2048 ? 1 : (unsigned)std::max(1, std::min(16, shn_host_cpus()));
    std::atomic<size_t> next{0};
    auto work = [&]() {
      std::string buf;
      for (size_t i0; (i0 = next.fetch_add(256)) < lines.size();)
        for (size_t i = i0; i < std::min(lines.size(), i0 + 256); i++) {
          const SV line = lines[i];
          size_t a = 0;
          while (a < line.size() && is_ws((uint8_t)line[a])) a++;
          if (a == line.size() || line[a] == '>' || line.size() <= 200) continue;
          const SV cur = strip(line);
          hf[i] = StringInterner::hash(cur.data(), cur.size());
          if (ds) { revcomp_into(cur, buf); hr[i] = StringInterner::hash(buf.data(), buf.size()); }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  }
  // (open-addressing tables: the node-based std containers were a cache miss per operation, six operations per record)
  StringInterner seen_names(lines.size() / 2 + 16);       // first token of a header line -> id; seen_count[id] = times met
  std::vector<uint64_t> seen_count;
  FlatMultiMap contigs(lines.size() / 2 + 16);            // hash of a kept sequence -> the lines that hold it (confirmed on the text)
  std::string rcbuf;
  auto kept_has = [&](uint64_t h, SV what) {
    for (int32_t v = contigs.find(h); v != -1; v = contigs.nxt(v)) if (strip(lines[(size_t)contigs.va(v)]) == what) return true;
    return false;
  };
  // "is the reverse complement of `cur` a kept sequence" without writing it out, eight bases at a time -- for sequences of A C G T
  // only (the fingerprint pass says which lines are): the complement of a letter is the letter ^ 0x15 (A <-> T) or ^ 0x04 (C <-> G),
  // and bit 1 tells the two pairs apart.  Half of the records of a double-stranded run are the reverse complements of earlier ones;
  // the byte-wise reverse complement of each was most of this loop.
  auto rc_equal = [](SV a, SV b) {
    const size_t L = a.size();
    if (b.size() != L) return false;
    size_t i = 0;
    for (; i + 8 <= L; i += 8) {
      uint64_t wa, wb;
      memcpy(&wa, a.data() + i, 8); memcpy(&wb, b.data() + L - 8 - i, 8);
      wb = __builtin_bswap64(wb);
      const uint64_t bit1 = (wb >> 1) & 0x0101010101010101ULL;
      if ((wb ^ 0x1515151515151515ULL ^ (bit1 * 0x11)) != wa) return false;
    }
    for (; i < L; i++) { const uint8_t c = (uint8_t)b[L - 1 - i]; const uint8_t cc = (uint8_t)(c ^ 0x15 ^ (((c >> 1) & 1) * 0x11)); if (cc != (uint8_t)a[i]) return false; }
    return true;
  };
  auto kept_has_rc = [&](uint64_t h, SV cur, size_t li) {
    bool all_fast = acgt && (*acgt)[li];
    if (all_fast)
      for (int32_t v = contigs.find(h); v != -1; v = contigs.nxt(v)) if (!(*acgt)[(size_t)contigs.va(v)]) { all_fast = false; break; }
    if (all_fast) {
      for (int32_t v = contigs.find(h); v != -1; v = contigs.nxt(v)) if (rc_equal(cur, strip(lines[(size_t)contigs.va(v)]))) return true;
      return false;
    }
    revcomp_into(cur, rcbuf);
    return kept_has(h, SV(rcbuf));
  };
  SV last;
  for (size_t li = 0; li < lines.size(); li++) {
    const SV line = lines[li];
    // (the table slots of the sequence lines a few records ahead on their way: two or three cache misses per record otherwise)
    if (li + 6 < lines.size() && lines[li + 6].size() > 200) { contigs.prefetch(hf[li + 6]); if (ds) contigs.prefetch(hr[li + 6]); }
    if (li + 12 < lines.size()) __builtin_prefetch(lines[li + 12].data());         // (a header line's first bytes: a record is ~1 KB of text further on)
    // tok = line.split()
    size_t a = 0;
    while (a < line.size() && is_ws((uint8_t)line[a])) a++;
    if (a == line.size()) return shn_fail(SHN_ERR_ARG, "shn_post_finalize: empty line");           // (tok[0] raises in the reference)
    if (line[a] == '>') {
      size_t b = a;
      while (b < line.size() && !is_ws((uint8_t)line[b])) b++;
      const SV tok0 = line.substr(a, b - a);
      bool is_new = false;
      const int32_t id = seen_names.intern(tok0.data(), tok0.size(), &is_new);
      if (!is_new) {
        std::string nl(tok0);
        nl += "_" + std::to_string(seen_count[(size_t)id]);
        bool first = true;
        for (size_t c = b; c < line.size();) {
          while (c < line.size() && is_ws((uint8_t)line[c])) c++;
          size_t d = c;
          while (d < line.size() && !is_ws((uint8_t)line[d])) d++;
          if (d > c) { if (!first) nl += "\t"; nl.append(line.data() + c, d - c); first = false; }
          c = d;
        }
        nl += "\n";
        seen_count[(size_t)id]++;
        own.push_back(std::move(nl));
        last = SV(own.back());
      } else { seen_count.push_back(1); last = line; }
    } else if (line.size() > 200) {
      const SV cur = strip(line);
      if (kept_has(hf[li], cur)) continue;
      if (ds && contigs.find(hr[li]) != -1 && kept_has_rc(hr[li], cur, li)) continue;
      contigs.add(hf[li], (int32_t)li);
      recs.push_back({last, line});
      rec_li.push_back((uint32_t)li);
    }
  }
  const bool dbgp = getenv("SHN_DEBUG") != nullptr || getenv("SHN_POST_LAPS") != nullptr;
  auto nowp = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
  const double tp0 = nowp();
  // ---- length sort: a dict keyed by header line (the last sequence of a repeated header line wins), by (len(seq line), header line)
  StringInterner by_header(recs.size() + 16);           // header line -> position in `order` (ids are given in order of first occurrence)
  std::vector<uint32_t> order;
  for (uint32_t i = 0; i < recs.size(); i++) {
    bool is_new = false;
    const int32_t id = by_header.intern(recs[i].first.data(), recs[i].first.size(), &is_new);
    if (is_new) order.push_back(i);
    else order[(size_t)id] = i;
  }
  // by (length of the sequence line, header line): the lengths as integers first, the header lines -- long common prefixes -- only
  // inside a group of one length (a comparison sort over both was 3.8 M string comparisons at the 222 k records of --config 2p);
  // the groups on the host threads
  {
    std::vector<uint64_t> by_len(order.size());
    for (size_t i = 0; i < order.size(); i++) by_len[i] = ((uint64_t)recs[order[i]].second.size() << 32) | (uint64_t)i;
    std::sort(by_len.begin(), by_len.end());
    std::vector<uint32_t> sorted(order.size());
    std::vector<size_t> group_at;
    for (size_t i = 0; i < by_len.size(); i++) {
      sorted[i] = order[(size_t)(uint32_t)by_len[i]];
      if (i == 0 || (by_len[i] >> 32) != (by_len[i - 1] >> 32)) group_at.push_back(i);
    }
    group_at.push_back(by_len.size());
    order.swap(sorted);
    auto groups = [&](size_t g0, size_t g1) {
      for (size_t g = g0; g < g1; g++)
        if (group_at[g + 1] - group_at[g] > 1)
          std::sort(order.begin() + (ptrdiff_t)group_at[g], order.begin() + (ptrdiff_t)group_at[g + 1], [&](uint32_t x, uint32_t y) { return recs[x].first < recs[y].first; });
    };
    const size_t ng = group_at.size() - 1;
    const unsigned nt = order.size() < 16384 ? 1u : (unsigned)std::max(1, std::min(16, shn_host_cpus()));
    if (nt <= 1) groups(0, ng);
    else {
      // (slices of about the same number of records)
      std::vector<std::thread> th;
      size_t g0 = 0;
      for (unsigned t = 0; t < nt; t++) {
        const size_t want = order.size() * (t + 1) / nt;
        size_t g1 = g0;
        while (g1 < ng && group_at[g1 + 1] <= want) g1++;
        if (t + 1 == nt) g1 = ng;
        if (g1 > g0) th.emplace_back(groups, g0, g1);
        g0 = g1;
      }
      for (auto& x : th) x.join();
    }
  }
  // ---- find_reps over (name = first field without '>', stripped sequence)
  const uint64_t n = order.size();
  std::vector<const uint8_t*> nptr(n), sptr(n);
  std::vector<uint64_t> nlen(n), slen(n), sgoff(n);
  for (uint64_t i = 0; i < n; i++) {
    const SV h = strip(recs[order[i]].first);
    size_t b = 0;
    while (b < h.size() && !is_ws((uint8_t)h[b])) b++;
    nptr[i] = (const uint8_t*)h.data() + 1; nlen[i] = b - 1;
    const SV ln = recs[order[i]].second;
    const SV sq = strip(ln);
    sptr[i] = (const uint8_t*)sq.data(); slen[i] = sq.size();
    sgoff[i] = line_goff[rec_li[order[i]]] + (uint64_t)(sq.data() - ln.data());
  }
  std::vector<uint8_t> keep(n + 1, 0);
  const double tp1 = nowp();
  if (n) {
    // (faster_reps.py is always run with -d, whatever the strandedness of the run: shannon.py:604; `ds` is the user's flag, which
    // process_concatenated_fasta.py gets, :596)
    int rc = find_reps_core(nptr.data(), nlen.data(), sptr.data(), slen.data(), n, 1, r, keep.data(), dev, sgoff.data());
    if (rc) return rc;
  }
  shn_post* o = new shn_post();
  for (uint64_t i = 0; i < n; i++) {
    if (!keep[i]) continue;
    o->names.append((const char*)nptr[i], nlen[i]);
    o->seqs.append((const char*)sptr[i], slen[i]);
    o->name_off.push_back(o->names.size());
    o->seq_off.push_back(o->seqs.size());
  }
  *out = o;
  if (dbgp) fprintf(stderr, "[post] lines %.3f s (from after the newline count), sort+pack %.3f s, find_reps %.3f s, %llu records\n", tp0 - tq0, tp1 - tp0, nowp() - tp1, (unsigned long long)n);
  return SHN_OK;
}
