// K-mer seed scans of reads against graph nodes (rows a17, a21): the inner loops of
// Read.find_bridging_reads (mbgraph.py:88-111: every read x every offset 1..len-K-1 probed against the
// first K bases of the X-nodes) and of known_paths (mbgraph.py:1355-1388: first and last K-mer of every
// read probed against the index of all K-mers of all nodes).  The string comparisons that follow a
// seed hit (Read.bridges, search_sequence) stay on the host: hits are few.
#include "common.h"
#include <cstring>
#include <algorithm>

#define SBLK2 256

struct SView { const uint64_t* words; const uint64_t* woff; const uint32_t* len; uint64_t n; uint32_t fixed_len, wpr; };

// mode 0, interior starts only (find_bridging_reads): ONE thread per read with a rolling K-mer.  The patterns are few (the K-mers
// at the start of the graph's x-nodes: 10^2-10^4 per partition) and the windows many (74 per 100-base read), so in front of the
// table stands a 2^18-bit filter held in the LDS: 98-99 % of the windows end at one LDS read.  A read's hits are written one after
// the other from the read's own offset: (read, start) order without ballots; the FILL pass skips the reads without a hit.
// (round 3's form -- a thread per window, two passes of table look-ups -- was 83 ms per step at BASELINE configs[2])
#define SEED_BM_BITS 18
#define SEED_BM_WORDS (1u << (SEED_BM_BITS - 5))
__device__ __forceinline__ uint32_t seed_bm_hash(uint64_t key) {
  const uint32_t x = ((uint32_t)key * 0x9E3779B1u) ^ ((uint32_t)(key >> 32) * 0x85EBCA6Bu);
  return x >> (32 - SEED_BM_BITS);
}
__global__ void seed_bm_build_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ bm) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t h = seed_bm_hash(keys[i]);
  atomicOr(&bm[h >> 5], 1u << (h & 31));
}
template <bool FILL>
__global__ __launch_bounds__(SBLK2) void seed_scan_reads_kernel(SView v, int K, const uint32_t* __restrict__ bm_g, const uint64_t* __restrict__ tkeys,
                                                                const uint32_t* __restrict__ tvals, const uint64_t* __restrict__ boff, int bits,
                                                                uint32_t* __restrict__ counts, const uint64_t* __restrict__ offs,
                                                                uint32_t* __restrict__ o_read, uint32_t* __restrict__ o_start, uint32_t* __restrict__ o_id) {
  __shared__ uint32_t bm[SEED_BM_WORDS];
  for (uint32_t i = threadIdx.x; i < SEED_BM_WORDS; i += SBLK2) bm[i] = bm_g[i];
  __syncthreads();
  const uint64_t mask = K == 32 ? ~0ULL : ((1ULL << (2 * K)) - 1);
  for (uint64_t r = (uint64_t)blockIdx.x * SBLK2 + threadIdx.x; r < v.n; r += (uint64_t)gridDim.x * SBLK2) {
    uint64_t o = 0;
    if (FILL) { o = offs[r]; if (offs[r + 1] == o) continue; }
    const uint32_t len = v.len ? v.len[r] : v.fixed_len;
    uint32_t c = 0;
    if (len >= (uint32_t)K + 2) {                                   // starts 1 .. len-K-1 (range(1, len - K))
      const uint64_t* __restrict__ w = v.words + (v.woff ? v.woff[r] : r * v.wpr);
      uint64_t key = shn_extract(w, 1, K);
      uint32_t p = (uint32_t)K + 1;                                  // the base the next start brings in
      uint64_t cur = w[p >> 5];
      const uint32_t last = len - (uint32_t)K - 1;
      for (uint32_t start = 1; start <= last; start++) {
        const uint32_t h = seed_bm_hash(key);
        if ((bm[h >> 5] >> (h & 31)) & 1u) {
          const int64_t j = shn_table_find(tkeys, boff, bits, key);
          const uint32_t id = j >= 0 ? tvals[j] : 0u;
          if (id) {
            if (FILL) { o_read[o + c] = (uint32_t)r; o_start[o + c] = start; o_id[o + c] = id - 1; }
            c++;
          }
        }
        if (start < last) {
          key = ((key << 2) | ((cur >> (62 - 2 * (p & 31))) & 3ULL)) & mask;
          p++;
          if ((p & 31) == 0) cur = w[p >> 5];
        }
      }
    }
    if (!FILL) counts[r] = c;
  }
}

// mode 1: first / last K-mer of every read
__global__ void seed_ends_kernel(SView v, int K, const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ tvals,
                                 const uint64_t* __restrict__ boff, int bits, uint32_t* __restrict__ first_id,
                                 uint32_t* __restrict__ last_id) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= v.n) return;
  uint32_t len = v.len ? v.len[r] : v.fixed_len;
  uint32_t a = 0, b = 0;
  if (len >= (uint32_t)K) {
    uint64_t wb = v.woff ? v.woff[r] : r * v.wpr;
    int64_t j = shn_table_find(tkeys, boff, bits, shn_extract(v.words + wb, 0, K));
    if (j >= 0) a = tvals[j];
    j = shn_table_find(tkeys, boff, bits, shn_extract(v.words + wb, len - K, K));
    if (j >= 0) b = tvals[j];
  }
  first_id[r] = a;
  last_id[r] = b;
}

static SView sview(const shn_reads* r) {
  SView v; v.words = r->d_words; v.woff = r->d_woff; v.len = r->d_len; v.n = r->n_reads; v.fixed_len = r->fixed_len; v.wpr = r->wpr;
  return v;
}

// Interior seed hits: every read, every start in [1, len-K): pattern id (table value - 1) if the K-mer is a
// key of `patterns`.  Call with out_read == NULL to get *n_hits, then with arrays of that size.
static int seed_scan_impl(shn_ctx* ctx, const shn_reads* reads, int K, const shn_table* patterns, uint64_t* n_hits,
                          uint32_t* out_read, uint32_t* out_start, uint32_t* out_id) {
  if (!ctx || !reads || !patterns || !n_hits) return shn_fail(SHN_ERR_ARG, "shn_seed_scan: NULL argument");
  if (reads->n_invalid) return shn_fail(SHN_ERR_ARG, "shn_seed_scan: reads contain non-ACGT bases");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_SEEDS);
  SView v = sview(reads);
  uint32_t max_win = reads->max_len > (uint32_t)K + 1 ? reads->max_len - K - 1 : 0;
  uint64_t total = v.n * max_win;
  if (total == 0) { *n_hits = 0; return SHN_OK; }
  int rc;
  // a thread per read behind an LDS filter (seed_scan_reads_kernel)
  const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(v.n, SBLK2), 4096);
  const uint64_t n_cnt = v.n;                                   // counts: per read
  // The call that asks for the number of hits and the one that fetches them come in pairs: the second finds the offsets (and the
  // filter) of the first -- same reads, same patterns, this thread.  They live in two blocks of the caching allocator that belong to
  // the PAIR (handed out by the first call, given back by the second, or by the next first call of this thread when no second
  // came): until round 5 they lay in the context's workspace slots cws[1..3], where any other call on the context between the two
  // could overwrite them silently (SHN_DEV_POISON_WS=1 crashed there).
  struct Pair { const shn_ctx* ctx; const shn_reads* reads; const shn_table* pat; uint64_t total, n_reads, n_pat, nh; int K;
                uint64_t* d_off; uint32_t* d_bm; hipStream_t stream; };
  static thread_local Pair last = {};
  auto drop = [&]() {
    if (last.d_off) shn_dev_free_on(last.d_off, last.stream);
    if (last.d_bm) shn_dev_free_on(last.d_bm, last.stream);
    last = {};
  };
  const bool second = out_read && last.ctx == ctx && last.reads == reads && last.pat == patterns && last.total == total &&
                      last.n_reads == reads->n_reads && last.n_pat == patterns->n && last.K == K && last.stream == s;
  uint64_t nh = 0;
  if (second) nh = last.nh;
  else {
    drop();
    ShnDevBufs tmp(s);
    uint32_t* pc = nullptr;
    HIP_TRY(tmp.get(&pc, (n_cnt + 1) * 4));
    HIP_TRY(shn_dev_malloc_on((void**)&last.d_off, (n_cnt + 2) * 8, s));
    last.stream = s;
    { hipError_t e = shn_dev_malloc_on((void**)&last.d_bm, SEED_BM_WORDS * 4, s);
      if (e != hipSuccess) { drop(); return shn_fail(SHN_ERR_HIP, std::string("shn_seed_scan: ") + hipGetErrorString(e)); } }
    hipError_t e = hipMemsetAsync(last.d_bm, 0, SEED_BM_WORDS * 4, s);
    if (e != hipSuccess) { drop(); return shn_fail(SHN_ERR_HIP, std::string("shn_seed_scan: ") + hipGetErrorString(e)); }
    if (patterns->n) hipLaunchKernelGGL(seed_bm_build_kernel, dim3((uint32_t)cdiv(patterns->n, 256)), dim3(256), 0, s, patterns->d_keys, patterns->n, last.d_bm);
    { TimerRegion tsc(ctx, T_SEED_SCAN); tsc.bytes(v.n * ((uint64_t)reads->wpr * 8 + 4));            // a read's packed words + its hit count
      hipLaunchKernelGGL((seed_scan_reads_kernel<false>), dim3(grid), dim3(SBLK2), 0, s, v, K, (const uint32_t*)last.d_bm, patterns->d_keys, patterns->d_counts,
                         patterns->d_bucket_off, patterns->bits, pc, nullptr, nullptr, nullptr, nullptr); }
    if ((rc = shn_device_scan_u32(ctx, pc, n_cnt, last.d_off, &nh))) { drop(); return rc; }      // (synchronises: pc may go back)
  }
  *n_hits = nh;
  if (!out_read) {
    if (nh == 0) { drop(); return SHN_OK; }                       // (nothing to fetch: no second call will come)
    last.ctx = ctx; last.reads = reads; last.pat = patterns; last.total = total; last.n_reads = reads->n_reads; last.n_pat = patterns->n;
    last.nh = nh; last.K = K;
    return SHN_OK;
  }
  struct DropAtExit { decltype(drop)& d; ~DropAtExit() { d(); } } at_exit{drop};      // the pair ends here, whatever happens below
  if (nh == 0) return SHN_OK;
  uint32_t *d_r = nullptr, *d_s = nullptr, *d_i = nullptr;       // (from the caching allocator: this runs once per partition, under the GPU mutex)
  ShnDevBufs hb(ctx->stream);
  HIP_TRY(hb.get(&d_r, nh * 4)); HIP_TRY(hb.get(&d_s, nh * 4)); HIP_TRY(hb.get(&d_i, nh * 4));
  { TimerRegion tsc(ctx, T_SEED_SCAN); tsc.bytes(v.n * ((uint64_t)reads->wpr * 8 + 8) + nh * 12);         // the words again, the read's offset, its hits written (read, start, pattern)
    hipLaunchKernelGGL((seed_scan_reads_kernel<true>), dim3(grid), dim3(SBLK2), 0, s, v, K, (const uint32_t*)last.d_bm, patterns->d_keys, patterns->d_counts,
                       patterns->d_bucket_off, patterns->bits, nullptr, (const uint64_t*)last.d_off, d_r, d_s, d_i); }
  HIP_TRY(hipMemcpyAsync(out_read, d_r, nh * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_start, d_s, nh * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(out_id, d_i, nh * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

extern "C" int shn_seed_scan(shn_ctx* ctx, const shn_reads* reads, int K, const shn_table* patterns, uint64_t* n_hits,
                             uint32_t* out_read, uint32_t* out_start, uint32_t* out_id) {
  return seed_scan_impl(ctx, reads, K, patterns, n_hits, out_read, out_start, out_id);
}
// first_id[r] / last_id[r] = table value (0 = absent) of the first / last K-mer of read r
extern "C" int shn_seed_ends(shn_ctx* ctx, const shn_reads* reads, int K, const shn_table* patterns, uint32_t* first_id,
                             uint32_t* last_id) {
  if (!ctx || !reads || !patterns || !first_id || !last_id) return shn_fail(SHN_ERR_ARG, "shn_seed_ends: NULL argument");
  if (reads->n_invalid) return shn_fail(SHN_ERR_ARG, "shn_seed_ends: reads contain non-ACGT bases");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_SEEDS);
  if (reads->n_reads == 0) return SHN_OK;
  SView v = sview(reads);
  uint32_t *d_a = nullptr, *d_b = nullptr;
  ShnDevBufs hb(ctx->stream);
  HIP_TRY(hb.get(&d_a, v.n * 4)); HIP_TRY(hb.get(&d_b, v.n * 4));
  hipLaunchKernelGGL(seed_ends_kernel, dim3((uint32_t)cdiv(v.n, 256)), dim3(256), 0, s, v, K, patterns->d_keys, patterns->d_counts,
                     patterns->d_bucket_off, patterns->bits, d_a, d_b);
  HIP_TRY(hipMemcpyAsync(first_id, d_a, v.n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(last_id, d_b, v.n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

// ------------------------------------------------------------------------------------------------------------
// r-mer join of two sets of sequences: every window of every sequence of `cands` against every occurrence of the same
// r-mer in the sequences of `foreign` -> (candidate, window start, foreign sequence) triples.  This feeds the guard of the
// component-sharded duplicate check.  Device: window keys of the foreign sequences sorted by key, every
// candidate window binary-searches its run.
__global__ void fs_keys_kernel(SView v, int K, uint32_t max_win, uint64_t* __restrict__ keys, uint32_t* __restrict__ ids,
                               const uint64_t* __restrict__ offs, uint32_t* __restrict__ counts) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= v.n * max_win) return;
  uint64_t r = gid / max_win;
  uint32_t start = (uint32_t)(gid - r * max_win);
  uint32_t len = v.len ? v.len[r] : v.fixed_len;
  bool have = len >= (uint32_t)K && start <= len - K;
  if (counts) { counts[gid] = have ? 1u : 0u; return; }
  if (!have) return;
  uint64_t wb = v.woff ? v.woff[r] : r * v.wpr;
  uint64_t o = offs[gid];
  keys[o] = shn_extract(v.words + wb, start, K);
  ids[o] = (uint32_t)r;
}
__device__ __forceinline__ uint64_t fs_lower_bound(const uint64_t* __restrict__ a, uint64_t n, uint64_t key) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
// FILL = false: counts[gid] = number of foreign occurrences of this candidate window; true: write the triples
template <bool FILL>
__global__ void fs_join_kernel(SView v, int K, uint32_t max_win, const uint64_t* __restrict__ fkeys, const uint32_t* __restrict__ fids,
                               uint64_t nf, uint32_t* __restrict__ counts, const uint64_t* __restrict__ offs, uint32_t* __restrict__ o_cand,
                               uint32_t* __restrict__ o_start, uint32_t* __restrict__ o_fid) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= v.n * max_win) return;
  uint64_t r = gid / max_win;
  uint32_t start = (uint32_t)(gid - r * max_win);
  uint32_t len = v.len ? v.len[r] : v.fixed_len;
  uint32_t c = 0;
  if (len >= (uint32_t)K && start <= len - K) {
    uint64_t wb = v.woff ? v.woff[r] : r * v.wpr;
    uint64_t key = shn_extract(v.words + wb, start, K);
    uint64_t lo = fs_lower_bound(fkeys, nf, key);
    uint64_t o = FILL ? offs[gid] : 0;
    for (uint64_t j = lo; j < nf && fkeys[j] == key; j++) {
      if (FILL) { o_cand[o + c] = (uint32_t)r; o_start[o + c] = start; o_fid[o + c] = fids[j]; }
      c++;
    }
  }
  if (!FILL) counts[gid] = c;
}

// Call with out_cand == NULL for *n_hits, then with arrays of that size.  Triples come in (candidate, start) order.
extern "C" int shn_rmer_join(shn_ctx* ctx, const shn_reads* cands, const shn_reads* foreign, int r, uint64_t* n_hits,
                             uint32_t* out_cand, uint32_t* out_start, uint32_t* out_foreign) {
  if (!ctx || !cands || !foreign || !n_hits) return shn_fail(SHN_ERR_ARG, "shn_rmer_join: NULL argument");
  if (r < 1 || r > 32) return shn_fail(SHN_ERR_ARG, "shn_rmer_join: r must be in [1,32]");
  if (cands->n_invalid || foreign->n_invalid) return shn_fail(SHN_ERR_ARG, "shn_rmer_join: non-ACGT bases");
  *n_hits = 0;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_SEEDS);
  SView vf = sview(foreign), vc = sview(cands);
  const uint32_t wf = foreign->max_len >= (uint32_t)r ? foreign->max_len - r + 1 : 0, wc = cands->max_len >= (uint32_t)r ? cands->max_len - r + 1 : 0;
  const uint64_t tf = vf.n * wf, tc = vc.n * wc;
  if (!tf || !tc) return SHN_OK;
  int rc;
  void *pc, *po;
  if ((rc = shn_ws(ctx)[25].get((std::max(tf, tc) + 1) * 4, &pc)) || (rc = shn_ws(ctx)[26].get((std::max(tf, tc) + 2) * 8, &po))) return rc;
  uint32_t* d_cnt = (uint32_t*)pc; uint64_t* d_off = (uint64_t*)po;
  hipLaunchKernelGGL(fs_keys_kernel, dim3((uint32_t)cdiv(tf, SBLK2)), dim3(SBLK2), 0, s, vf, r, wf, nullptr, nullptr, nullptr, d_cnt);
  uint64_t nf = 0;
  if ((rc = shn_device_scan_u32(ctx, d_cnt, tf, d_off, &nf))) return rc;
  if (!nf) return SHN_OK;
  uint64_t *fk = nullptr, *fk2 = nullptr; uint32_t *fi = nullptr, *fi2 = nullptr, *oc = nullptr, *os = nullptr, *of = nullptr;
  auto cleanup = [&]() { shn_dev_free(fk); shn_dev_free(fk2); shn_dev_free(fi); shn_dev_free(fi2); shn_dev_free(oc); shn_dev_free(os); shn_dev_free(of); };
  if (shn_dev_malloc(&fk, nf * 8) != hipSuccess || shn_dev_malloc(&fk2, nf * 8) != hipSuccess || shn_dev_malloc(&fi, nf * 4) != hipSuccess ||
      shn_dev_malloc(&fi2, nf * 4) != hipSuccess) { cleanup(); return shn_fail(SHN_ERR_NOMEM, "shn_rmer_join: out of device memory"); }
  hipLaunchKernelGGL(fs_keys_kernel, dim3((uint32_t)cdiv(tf, SBLK2)), dim3(SBLK2), 0, s, vf, r, wf, fk, fi, (const uint64_t*)d_off, nullptr);
  if ((rc = shn_sort_pairs(ctx, fk, fi, fk2, fi2, nf, 0, 2 * r))) { cleanup(); return rc; }
  hipLaunchKernelGGL(fs_join_kernel<false>, dim3((uint32_t)cdiv(tc, SBLK2)), dim3(SBLK2), 0, s, vc, r, wc, fk, fi, nf, d_cnt, nullptr, nullptr, nullptr, nullptr);
  uint64_t nh = 0;
  if ((rc = shn_device_scan_u32(ctx, d_cnt, tc, d_off, &nh))) { cleanup(); return rc; }
  *n_hits = nh;
  if (!out_cand || !nh) { cleanup(); return SHN_OK; }
  if (shn_dev_malloc(&oc, nh * 4) != hipSuccess || shn_dev_malloc(&os, nh * 4) != hipSuccess || shn_dev_malloc(&of, nh * 4) != hipSuccess) {
    cleanup(); return shn_fail(SHN_ERR_NOMEM, "shn_rmer_join: out of device memory"); }
  hipLaunchKernelGGL(fs_join_kernel<true>, dim3((uint32_t)cdiv(tc, SBLK2)), dim3(SBLK2), 0, s, vc, r, wc, fk, fi, nf, nullptr, (const uint64_t*)d_off, oc, os, of);
  hipError_t e = hipMemcpyAsync(out_cand, oc, nh * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(out_start, os, nh * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(out_foreign, of, nh * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cleanup();
  if (e != hipSuccess) return shn_fail(SHN_ERR_HIP, hipGetErrorString(e));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}
