// Read files straight into HBM: SURVEY §8 row f2 (the reference reads its FASTA line by line in Python -- rc_gnu.py:15-20
// find_L takes every second line as a read, kmers_for_component.py:329-403 and multibridging.py:185-236 do the same -- and
// converts FASTQ to that form first).
//
// shn_reads_ingest takes the text of a read file (2-line FASTA records "name line, sequence line", or 4-line FASTQ records) and
// leaves the reads 2-bit packed on the device, and optionally as a host code matrix (what ReadStore / shn_mbgraph_run_rows read
// the graph stage's read text from):
//   pass 1  host threads scan byte ranges of the text (cut at record starts) for newlines: records and read length per range
//   pass 2  groups of ranges are parsed into one of two pinned staging buffers (codes 0..3, 4 = anything else) while the
//           previous group is on its way: hipMemcpyAsync -> pack_kernel into the read set's final arrays, one launch per group
// shn_reads_ingest takes reads of one length (fixed-length read sets are what the routing / graph stages address by row);
// shn_reads_ingest_ragged takes reads of any lengths (the reference's own Samples/SE_read.fasta has 48-51 bases per read): the same
// scan and the same threaded parse into ONE flat code array with per-read offsets, packed by the ragged path of shn_reads_create.
// Bases outside ACGT are kept (code 4: the packed set's mask / bad-read flags); multi-line FASTA or a malformed record is refused
// with SHN_ERR_ARG and a message starting "shn_reads_ingest: unsupported", and the caller reads the file record by record.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

int shn_pack_fixed_codes(shn_ctx* ctx, const uint8_t* d_codes, uint64_t n, uint32_t L, uint32_t wpr, uint64_t* d_words, uint64_t* d_mask,
                         hipStream_t s);
int shn_reads_finish_fixed(shn_ctx* ctx, shn_reads* r);

namespace {

struct Range {
  uint64_t b0 = 0, b1 = 0, n_rec = 0, rec0 = 0;
  uint32_t min_len = 0xFFFFFFFFu, max_len = 0;
  int bad = 0;            // 1: a record does not start with the record character, 2: truncated record
};

inline uint64_t line_end(const uint8_t* t, uint64_t n, uint64_t p) {       // index of the '\n' ending the line at p (n if none)
  const void* q = memchr(t + p, '\n', n - p);
  return q ? (uint64_t)((const uint8_t*)q - t) : n;
}

// first record start at or after pos
uint64_t align_record(const uint8_t* t, uint64_t n, uint64_t pos, bool fastq) {
  if (pos == 0) return 0;
  uint64_t q = pos;
  if (t[pos - 1] != '\n') { q = line_end(t, n, pos); if (q >= n) return n; q++; }
  while (q < n) {
    if (!fastq) { if (t[q] == '>') return q; }
    else if (t[q] == '@') {                                 // a quality line may start with '@' too: the line after next must be "+..."
      uint64_t e1 = line_end(t, n, q);
      if (e1 < n) { uint64_t e2 = line_end(t, n, e1 + 1); if (e2 + 1 < n && t[e2 + 1] == '+') return q; }
    }
    q = line_end(t, n, q);
    if (q >= n) return n;
    q++;
  }
  return n;
}

// visits the records of [b0, b1): fn(seq_start, seq_len); returns false on a malformed record
template <class F> int walk_records(const uint8_t* t, uint64_t n, uint64_t b0, uint64_t b1, bool fastq, F&& fn) {
  uint64_t p = b0;
  const uint8_t mark = fastq ? '@' : '>';
  while (p < b1) {
    if (t[p] != mark) {
      if (t[p] == '\n' || t[p] == '\r') { p++; continue; }        // blank lines between records (and at the end of the file)
      return 1;
    }
    uint64_t e = line_end(t, n, p);
    if (e >= n) return 2;
    const uint64_t s = e + 1;
    e = line_end(t, n, s);
    uint64_t len = e - s;
    if (len && t[s + len - 1] == '\r') len--;
    fn(s, len);
    p = e + 1;
    if (fastq) {
      if (p >= n || t[p] != '+') return 2;
      e = line_end(t, n, p);
      if (e >= n) return 2;
      e = line_end(t, n, e + 1);
      p = e + 1;
    }
  }
  return 0;
}

}  // namespace

// ---- a read file shared out by BYTES (the N-rank CLI, round 6): every rank looks at its own share of the text only.  Rank r counts
// the records that start in bytes [B r / W, B (r + 1) / W) (shn_text_records_in_range: the same resynchronisation on the record
// character as the threaded scan below uses for its ranges); the counts of all ranks give every share's first record number; the
// records [n r / W, n (r + 1) / W) a rank is to hold start a few records into one of the shares (shn_text_skip_records) and that
// stretch of text goes through shn_reads_ingest.  The reference streams the file once (kmers_for_component.py:322-403, 10 M reads at
// a time); until round 6 every rank here parsed the WHOLE file into a host matrix and kept a slice of it.
static bool text_is_fastq(const uint8_t* text, uint64_t n_bytes, int format) {
  if (format == 2) return true;
  if (format == 1) return false;
  uint64_t p = 0;
  while (p < n_bytes && (text[p] == '\n' || text[p] == '\r')) p++;
  return p < n_bytes && text[p] == '@';
}
extern "C" int shn_text_records_in_range(const uint8_t* text, uint64_t n_bytes, uint64_t lo, uint64_t hi, int format, uint64_t* first_out, uint64_t* n_records_out,
                                         uint64_t stride, uint64_t* index_out, uint64_t index_cap, uint64_t* n_index_out) {
  if ((n_bytes && !text) || !first_out || !n_records_out || lo > hi || hi > n_bytes || (index_out && (!stride || !n_index_out)))
    return shn_fail(SHN_ERR_ARG, "shn_text_records_in_range: bad argument");
  uint64_t n_idx = 0;
  const bool fastq = text_is_fastq(text, n_bytes, format);
  const uint64_t first = align_record(text, n_bytes, lo, fastq);
  uint64_t cnt = 0;
  if (first < hi) {
    // (the records that START before hi: the walk goes record by record and stops at the first start at or beyond hi)
    uint64_t p = first;
    const uint8_t mark = fastq ? '@' : '>';
    while (p < hi) {
      if (text[p] != mark) {
        if (text[p] == '\n' || text[p] == '\r') { p++; continue; }
        return shn_fail(SHN_ERR_ARG, "shn_text_records_in_range: unsupported: a record does not start with its record character");
      }
      if (index_out && cnt % stride == 0) { if (n_idx >= index_cap) return shn_fail(SHN_ERR_ARG, "shn_text_records_in_range: index_out too small"); index_out[n_idx++] = p; }
      cnt++;
      uint64_t e = line_end(text, n_bytes, p);                 // name line
      if (e >= n_bytes) break;
      e = line_end(text, n_bytes, e + 1);                      // sequence line
      p = e + 1;
      if (fastq && p < n_bytes) {
        e = line_end(text, n_bytes, p);                        // "+" line
        if (e >= n_bytes) break;
        e = line_end(text, n_bytes, e + 1);                    // quality line
        p = e + 1;
      }
    }
  }
  *first_out = first < hi ? first : n_bytes;
  *n_records_out = cnt;
  if (n_index_out) *n_index_out = n_idx;
  return SHN_OK;
}
// the start of the k-th record after the record that starts at `from` (k = 0: `from` itself); n_bytes when the text has fewer
extern "C" int shn_text_skip_records(const uint8_t* text, uint64_t n_bytes, uint64_t from, uint64_t k, int format, uint64_t* offset_out) {
  if ((n_bytes && !text) || !offset_out || from > n_bytes) return shn_fail(SHN_ERR_ARG, "shn_text_skip_records: bad argument");
  const bool fastq = text_is_fastq(text, n_bytes, format);
  const uint8_t mark = fastq ? '@' : '>';
  uint64_t p = from;
  while (k && p < n_bytes) {
    if (text[p] != mark) {
      if (text[p] == '\n' || text[p] == '\r') { p++; continue; }
      return shn_fail(SHN_ERR_ARG, "shn_text_skip_records: unsupported: a record does not start with its record character");
    }
    uint64_t e = line_end(text, n_bytes, p);
    if (e >= n_bytes) { p = n_bytes; break; }
    e = line_end(text, n_bytes, e + 1);
    p = std::min(n_bytes, e + 1);
    if (fastq && p < n_bytes) {
      e = line_end(text, n_bytes, p);
      if (e >= n_bytes) { p = n_bytes; break; }
      e = line_end(text, n_bytes, e + 1);
      p = std::min(n_bytes, e + 1);
    }
    k--;
  }
  while (p < n_bytes && (text[p] == '\n' || text[p] == '\r')) p++;       // (blank lines between records)
  *offset_out = k ? n_bytes : p;
  return SHN_OK;
}

// text / n_bytes: the file; format: 0 = by the first character ('>' FASTA, '@' FASTQ), 1 FASTA, 2 FASTQ; codes_out: NULL or room
// for codes_cap bytes, filled with the [n_reads][read length] code matrix (fails if too small: size it with a first call that
// passes out = NULL, which only scans); out: the packed read set (NULL: scan only).
extern "C" int shn_reads_ingest(shn_ctx* ctx, const uint8_t* text, uint64_t n_bytes, int format, uint8_t* codes_out, uint64_t codes_cap,
                                uint64_t* n_reads_out, uint32_t* read_len_out, shn_reads** out) {
  if ((n_bytes && !text) || !n_reads_out || !read_len_out || (out && !ctx)) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: NULL argument");
  *n_reads_out = 0; *read_len_out = 0;
  if (out) *out = nullptr;
  uint64_t lead = 0;
  while (lead < n_bytes && (text[lead] == '\n' || text[lead] == '\r')) lead++;
  if (lead >= n_bytes) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: unsupported: no records");
  if (format == 0) format = text[lead] == '@' ? 2 : 1;
  if (format != 1 && format != 2) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: bad format");
  const bool fastq = format == 2;
  const bool dbg = getenv("SHN_DEBUG") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_0 = now();
  double t_scan = 0, t_alloc = 0, t_parse = 0, t_wait = 0;
  const unsigned T = (unsigned)std::max(1, std::min(shn_host_cpus(), 64));
  // ranges of ~16 MB of text, cut at record starts
  uint64_t range_bytes = 16u << 20;
  if (const char* e = getenv("SHN_INGEST_RANGE_BYTES")) range_bytes = std::max<uint64_t>(1, strtoull(e, nullptr, 10));     // (tests: many small ranges)
  const uint64_t n_ranges = std::max<uint64_t>(1, std::min<uint64_t>(n_bytes / range_bytes + 1, 1u << 20));
  std::vector<Range> R(n_ranges);
  {
    std::atomic<uint64_t> next{0};
    auto cut = [&]() { for (uint64_t i; (i = next.fetch_add(1)) < n_ranges;) R[i].b0 = align_record(text, n_bytes, i == 0 ? lead : n_bytes / n_ranges * i, fastq); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<uint64_t>(T, n_ranges); t++) th.emplace_back(cut);
    cut();
    for (auto& x : th) x.join();
    for (uint64_t i = 0; i < n_ranges; i++) R[i].b1 = i + 1 < n_ranges ? R[i + 1].b0 : n_bytes;
  }
  {
    std::atomic<uint64_t> next{0};
    auto scan = [&]() {
      for (uint64_t i; (i = next.fetch_add(1)) < n_ranges;) {
        Range& r = R[i];
        if (r.b0 >= r.b1) continue;
        r.bad = walk_records(text, n_bytes, r.b0, r.b1, fastq, [&](uint64_t, uint64_t len) {
          r.n_rec++; r.min_len = std::min<uint32_t>(r.min_len, (uint32_t)std::min<uint64_t>(len, 0xFFFFFFFEu)); r.max_len = std::max<uint32_t>(r.max_len, (uint32_t)std::min<uint64_t>(len, 0xFFFFFFFEu)); });
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<uint64_t>(T, n_ranges); t++) th.emplace_back(scan);
    scan();
    for (auto& x : th) x.join();
  }
  uint64_t N = 0;
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  for (auto& r : R) {
    if (r.bad) return shn_fail(SHN_ERR_ARG, r.bad == 1 ? "shn_reads_ingest: unsupported: a record does not start with its marker (multi-line FASTA?)"
                                                       : "shn_reads_ingest: unsupported: truncated record");
    r.rec0 = N; N += r.n_rec;
    if (r.n_rec) { mn = std::min(mn, r.min_len); mx = std::max(mx, r.max_len); }
  }
  if (!N) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: unsupported: no records");
  if (mn != mx || mx == 0) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: unsupported: reads of different lengths (" + std::to_string(mn) + " .. " + std::to_string(mx) + ")");
  const uint32_t L = mx;
  *n_reads_out = N; *read_len_out = L;
  t_scan = now() - t_0;
  if (!out && !codes_out) return SHN_OK;
  if (codes_out && codes_cap < N * (uint64_t)L) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: codes_out too small");
  // base -> code without a table (a table look-up per base does not vectorise and was the bound of the whole ingest): bits 1-2 of
  // 'A' 'C' 'G' 'T' (either case) are 0 1 3 2; x ^ (x >> 1) puts G and T in order; anything else becomes 4
  auto encode = [](const uint8_t* __restrict__ p, uint8_t* __restrict__ d, uint32_t n) {
    for (uint32_t j = 0; j < n; j++) {
      const uint8_t c = p[j], u = (uint8_t)(c & 0xDF), t = (uint8_t)((c >> 1) & 3);
      const uint8_t ok = (uint8_t)((u == 'A') | (u == 'C') | (u == 'G') | (u == 'T'));
      d[j] = ok ? (uint8_t)(t ^ (t >> 1)) : (uint8_t)4;
    }
  };
  auto parse = [&](const Range& r, uint8_t* dst_a, uint8_t* dst_b) {             // dst_*: row 0 = record r.rec0 (either may be NULL)
    uint64_t k = 0;
    walk_records(text, n_bytes, r.b0, r.b1, fastq, [&](uint64_t s, uint64_t) {
      const uint8_t* p = text + s;
      if (dst_a) { uint8_t* d = dst_a + k * L; encode(p, d, L); if (dst_b) memcpy(dst_b + k * L, d, L); }
      else encode(p, dst_b + k * L, L);
      k++;
    });
  };
  if (!out) {                                                                   // host matrix only
    std::atomic<uint64_t> next{0};
    auto work = [&]() { for (uint64_t i; (i = next.fetch_add(1)) < n_ranges;) if (R[i].n_rec) parse(R[i], nullptr, codes_out + R[i].rec0 * L); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<uint64_t>(T, n_ranges); t++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    return SHN_OK;
  }
  SHN_ENTER(ctx);
  shn_reads* r = new shn_reads();
  memset(r, 0, sizeof(*r));
  r->ctx = ctx; r->device = ctx->device; r->n_reads = N; r->fixed_len = L; r->max_len = L;
  r->wpr = (uint32_t)(2 * cdiv(L, 64)); r->n_words = N * r->wpr; r->total_bases = N * (uint64_t)L;
  // staging: groups of consecutive ranges of at most `cap` records
  uint64_t cap = 0;
  for (auto& x : R) cap = std::max(cap, x.n_rec);
  uint64_t stage_bytes = 256ull << 20;
  if (const char* e = getenv("SHN_INGEST_STAGE_BYTES")) stage_bytes = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  cap = std::max<uint64_t>(cap, std::min<uint64_t>(N, stage_bytes / L + 1));
  // the pinned pair is kept between calls (pinning and unpinning 0.5 GB cost 70 of the 100 ms a 5 M-read file took); one ingest
  // at a time uses it
  static std::mutex pin_mu;
  static uint8_t* pin_keep[2] = {nullptr, nullptr};
  static size_t pin_cap = 0;
  std::lock_guard<std::mutex> pin_lock(pin_mu);
  uint8_t* pin[2] = {nullptr, nullptr};
  uint8_t* dst[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipStream_t s = ctx->stream; shn_use_stream(s);
  auto cleanup = [&]() {                       // (the staging blocks go back to the allocator only after the stream has drained)
    (void)hipStreamSynchronize(s);
    for (int b = 0; b < 2; b++) { if (dst[b]) shn_dev_free(dst[b]); if (ev[b]) hipEventDestroy(ev[b]); }
  };
#define TRYI(e) do { hipError_t _e = (e); if (_e != hipSuccess) { cleanup(); shn_reads_destroy(r); \
      return shn_fail(SHN_ERR_HIP, std::string("shn_reads_ingest: ") + #e + ": " + hipGetErrorString(_e)); } } while (0)
  TRYI(shn_hip_malloc(&r->d_words, (r->n_words + 2) * 8));
  TRYI(shn_hip_malloc(&r->d_mask, (r->n_words / 2 + 2) * 8));
  TRYI(hipMemsetAsync(r->d_words + r->n_words, 0, 16, s));
  TRYI(hipMemsetAsync(r->d_mask + r->n_words / 2, 0, 16, s));
  if (pin_cap < cap * L) {
    for (int b = 0; b < 2; b++) { if (pin_keep[b]) hipHostFree(pin_keep[b]); pin_keep[b] = nullptr; }
    pin_cap = 0;
    for (int b = 0; b < 2; b++) TRYI(hipHostMalloc(&pin_keep[b], cap * L, hipHostMallocDefault));
    pin_cap = cap * L;
  }
  for (int b = 0; b < 2; b++) {
    pin[b] = pin_keep[b];
    TRYI(shn_dev_malloc(&dst[b], cap * L));
    TRYI(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
  }
  t_alloc = now() - t_0 - t_scan;
  uint64_t i0 = 0;
  int b = 0;
  bool used[2] = {false, false};
  while (i0 < n_ranges) {
    uint64_t i1 = i0, n_grp = 0;
    while (i1 < n_ranges && n_grp + R[i1].n_rec <= cap) { n_grp += R[i1].n_rec; i1++; }
    if (n_grp) {
      { const double tw = now(); if (used[b]) TRYI(hipEventSynchronize(ev[b])); t_wait += now() - tw; }
      const uint64_t base = R[i0].rec0;
      const double tp = now();
      std::atomic<uint64_t> next{i0};
      auto work = [&]() {
        for (uint64_t i; (i = next.fetch_add(1)) < i1;)
          if (R[i].n_rec) parse(R[i], pin[b] + (R[i].rec0 - base) * L, codes_out ? codes_out + R[i].rec0 * L : nullptr);
      };
      std::vector<std::thread> th;
      for (unsigned t = 1; t < std::min<uint64_t>(T, i1 - i0); t++) th.emplace_back(work);
      work();
      for (auto& x : th) x.join();
      t_parse += now() - tp;
      TRYI(hipMemcpyAsync(dst[b], pin[b], n_grp * L, hipMemcpyHostToDevice, s));
      int rc = shn_pack_fixed_codes(ctx, dst[b], n_grp, L, r->wpr, r->d_words + base * r->wpr, r->d_mask + base * (r->wpr / 2), s);
      if (rc) { cleanup(); shn_reads_destroy(r); return rc; }
      TRYI(hipEventRecord(ev[b], s));
      used[b] = true;
      b ^= 1;
    }
    i0 = i1;
  }
  { const double tw = now(); TRYI(hipStreamSynchronize(s)); t_wait += now() - tw; }
#undef TRYI
  const double t_c = now();
  cleanup();
  if (dbg) fprintf(stderr, "[ingest] %llu reads of %u: scan %.3f s, allocations %.3f s, parse %.3f s, waiting for copies %.3f s, frees %.3f s, total %.3f s\n",
                   (unsigned long long)N, L, t_scan, t_alloc, t_parse, t_wait, now() - t_c, now() - t_0);
  int rc = shn_reads_finish_fixed(ctx, r);
  if (rc) { shn_reads_destroy(r); return rc; }
  *out = r;
  return SHN_OK;
}


// Reads of any lengths.  codes_out: room for codes_cap bytes (the text's size is always enough), filled with the reads' codes one
// after the other; offsets_out: n_reads + 1 entries (offsets_cap of them available).  out == NULL and codes_out == NULL: scan only
// (n_reads, max_len, total_bases).  Otherwise both codes_out and offsets_out are needed (the packed set is made from them).
extern "C" int shn_reads_ingest_ragged(shn_ctx* ctx, const uint8_t* text, uint64_t n_bytes, int format, uint8_t* codes_out, uint64_t codes_cap,
                                       uint64_t* offsets_out, uint64_t offsets_cap, uint64_t* n_reads_out, uint32_t* max_len_out,
                                       uint64_t* total_bases_out, shn_reads** out) {
  if ((n_bytes && !text) || !n_reads_out || !max_len_out || !total_bases_out || (out && !ctx))
    return shn_fail(SHN_ERR_ARG, "shn_reads_ingest_ragged: NULL argument");
  *n_reads_out = 0; *max_len_out = 0; *total_bases_out = 0;
  if (out) *out = nullptr;
  uint64_t lead = 0;
  while (lead < n_bytes && (text[lead] == '\n' || text[lead] == '\r')) lead++;
  if (lead >= n_bytes) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: unsupported: no records");
  if (format == 0) format = text[lead] == '@' ? 2 : 1;
  if (format != 1 && format != 2) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: bad format");
  const bool fastq = format == 2;
  const unsigned T = (unsigned)std::max(1, std::min(shn_host_cpus(), 64));
  uint64_t range_bytes = 16u << 20;
  if (const char* e = getenv("SHN_INGEST_RANGE_BYTES")) range_bytes = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
  const uint64_t n_ranges = std::max<uint64_t>(1, std::min<uint64_t>(n_bytes / range_bytes + 1, 1u << 20));
  struct RRange { uint64_t b0 = 0, b1 = 0, n_rec = 0, rec0 = 0, bases = 0, base0 = 0; uint32_t max_len = 0; int bad = 0; };
  std::vector<RRange> R(n_ranges);
  auto on_threads = [&](auto&& fn) {
    std::atomic<uint64_t> next{0};
    auto work = [&]() { for (uint64_t i; (i = next.fetch_add(1)) < n_ranges;) fn(i); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < std::min<uint64_t>(T, n_ranges); t++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  };
  on_threads([&](uint64_t i) { R[i].b0 = align_record(text, n_bytes, i == 0 ? lead : n_bytes / n_ranges * i, fastq); });
  for (uint64_t i = 0; i < n_ranges; i++) R[i].b1 = i + 1 < n_ranges ? R[i + 1].b0 : n_bytes;
  on_threads([&](uint64_t i) {
    RRange& r = R[i];
    if (r.b0 >= r.b1) return;
    r.bad = walk_records(text, n_bytes, r.b0, r.b1, fastq, [&](uint64_t, uint64_t len) {
      r.n_rec++; r.bases += len; r.max_len = std::max<uint32_t>(r.max_len, (uint32_t)std::min<uint64_t>(len, 0xFFFFFFFEu)); });
  });
  uint64_t N = 0, B = 0;
  uint32_t mx = 0;
  for (auto& r : R) {
    if (r.bad) return shn_fail(SHN_ERR_ARG, r.bad == 1 ? "shn_reads_ingest: unsupported: a record does not start with its marker (multi-line FASTA?)"
                                                       : "shn_reads_ingest: unsupported: truncated record");
    r.rec0 = N; r.base0 = B; N += r.n_rec; B += r.bases;
    mx = std::max(mx, r.max_len);
  }
  if (!N) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest: unsupported: no records");
  *n_reads_out = N; *max_len_out = mx; *total_bases_out = B;
  if (!out && !codes_out) return SHN_OK;
  if (!codes_out || !offsets_out) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest_ragged: codes_out and offsets_out are both needed");
  if (codes_cap < B || offsets_cap < N + 1) return shn_fail(SHN_ERR_ARG, "shn_reads_ingest_ragged: output arrays too small");
  on_threads([&](uint64_t i) {
    const RRange& r = R[i];
    if (!r.n_rec) return;
    uint64_t k = r.rec0, at = r.base0;
    walk_records(text, n_bytes, r.b0, r.b1, fastq, [&](uint64_t s, uint64_t len) {
      offsets_out[k++] = at;
      const uint8_t* p = text + s;
      uint8_t* d = codes_out + at;
      for (uint64_t j = 0; j < len; j++) {
        const uint8_t c = p[j], u = (uint8_t)(c & 0xDF), t = (uint8_t)((c >> 1) & 3);
        const uint8_t ok = (uint8_t)((u == 'A') | (u == 'C') | (u == 'G') | (u == 'T'));
        d[j] = ok ? (uint8_t)(t ^ (t >> 1)) : (uint8_t)4;
      }
      at += len;
    });
  });
  offsets_out[N] = B;
  if (!out) return SHN_OK;
  return shn_reads_create(ctx, codes_out, offsets_out, N, 0, SHN_ENC_CODES, out);
}
