// (K+1)-mer counting through super-k-mers (the path of large, diverse inputs; count.hip has the one-pass path in front of it and
// the partition pipeline behind it).  Replaces `jellyfish count | dump` (shannon.py:439-441) like the rest of count.hip.
//
// The partition pipeline moves every window of every read through HBM as an 8-byte key, twice written and three times read:
// ~3 KB per 100-base read where SURVEY 8d's model has 1.2 KB.  Here the windows travel as the reads' own 2-bit text: consecutive
// windows of a read that share their minimizer (the m-mer of smallest hash inside the window, m = 13; canonical, so both strands
// of a k1-mer pick the same one) are ONE 16-byte record -- up to 48 bases and the window count -- and all records of a minimizer
// land in the same bucket.  A 100-base read makes ~11 records (176 bytes) instead of 75 keys (600 bytes), and its windows are
// only expanded to keys inside the bucket kernel, in LDS:
//   sk_scan<false>   packed reads -> minimizers -> records per level-1 bucket (histogram only)
//   sk_scan<true>    the same pass, records staged in LDS and written to their level-1 partitions
//   skr_hist/scatter level 2: records to their final buckets (2^bits of them, ~3,500 windows each)
//   sk_buckets       one block per bucket: windows -> canonical keys -> LDS hash table -> (key, count) pairs; a table that fills
//                    up (a deeply covered locus drags its sequencing errors along: thousands of distinct keys) is written out
//                    and started again, so the pairs of a bucket may repeat a key
//   pairs path       shn_table_from_pairs reduces the pairs by key into the table every consumer expects (hash buckets, sorted)
// Every kernel is byte / integer work; nothing here is a contraction (no MFMA).
#include "count_views.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

#define SK_BLK 256
#define SK_RCAP 3072                                   // records staged in LDS per block of sk_scan<true>
#define SK_CHECK 4                                     // the staging buffer is looked at every SK_CHECK bases
#define SK_FLUSH_AT (SK_RCAP - (SK_CHECK + 1) * SK_BLK)   // a thread closes at most one record per base, and one at its read's end
#define SK_TILE 32768                                  // records per block tile of the level-2 kernels
#define SK_CAP 2048                                    // LDS hash slots of a bucket block (24 KB)
#ifndef SK_SCAP
#define SK_SCAP 2048                                   // ... of a block of the sorted-bucket kernel (layout 1)
#endif
#ifndef SK_STHREADS
#define SK_STHREADS 256                                // threads of such a block
#endif
#ifndef SK_T1CAP
#define SK_T1CAP 1024
#endif
#ifndef SK_T2THREADS
#define SK_T2THREADS 256
#endif
#define SK_SPILL 1280                                  // distinct keys at which the table is written out (one more round adds <= 32 x 18)
#define SK_EMPTY 0xFFFFFFFFFFFFFFFFULL
#define SK_REGIONS 256                                 // stretches of the pair pool, a cursor each (sk_buckets_sorted_kernel)
#define SK_RSTRIDE 16                                  // ... the cursors 128 bytes apart
#define SK_RANK_MAX 512                                // buckets of at most this many distinct keys are ordered by counting, larger ones by a bitonic sort
#define SK_MIN_K 20

struct SkParams { int k, m, w, bits, b1, b2; };

struct __attribute__((aligned(16))) SkRec { uint64_t hi; uint32_t lo; uint32_t meta; };   // bases 0..31, bases 32..47, windows | bucket << 5

#define sk_order shn_sk_order
#define sk_bucket shn_sk_bucket

// 96 bits of a read's 2-bit text from base p on (words beyond the read's own are taken as 0)
__device__ __forceinline__ void sk_extract96(const uint64_t* __restrict__ w, uint32_t nw, uint32_t p, uint64_t& hi, uint32_t& lo) {
  const uint32_t wi = p >> 5, sh = (p & 31) * 2;
  const uint64_t a = w[wi], b = wi + 1 < nw ? w[wi + 1] : 0ULL, c = (sh && wi + 2 < nw) ? w[wi + 2] : 0ULL;
  hi = sh ? (a << sh) | (b >> (64 - sh)) : a;
  const uint64_t l = sh ? (b << sh) | (c >> (64 - sh)) : b;
  lo = (uint32_t)(l >> 32);
}
// window j of a record: the 2k bits from base j on
__device__ __forceinline__ uint64_t sk_window(uint64_t hi, uint32_t lo, int j, int k) {
  const int sh = 2 * j;                                 // <= 34
  const uint64_t v = sh ? (hi << sh) | (((uint64_t)lo << 32) >> (64 - sh)) : hi;
  return v >> (64 - 2 * k);
}

// slot of a key in a bucket block's LDS table
__device__ __forceinline__ uint32_t sk_slot_hash(uint64_t key) { return (uint32_t)shn_mix64(key ^ 0x9E3779B97F4A7C15ULL); }

// ---------------------------------------------------------------- reads -> records
// One thread per read, one base per step: the m-mer ending at the base (forward and reverse complement rolled along) and its
// order value.  The minimum over the last w order values without a data-dependent rescan (a wavefront whose 64 reads rescan at
// different steps pays for a rescan at EVERY step): the m-mers are taken in blocks of w; `pref` is the running minimum of the
// current block, suf[t] the minimum of positions t.. of the block before (made by one backward pass when a block ends -- the
// same step in every lane); the window ending at position t of a block spans suf[t + 1] and pref.  A window is complete with its
// last m-mer; a run of at most w windows with the same minimizer is a record (2 k1 - m <= 48 bases).  Windows holding a base
// outside ACGT belong to no record (jellyfish / extension_correction count ACGT-only windows).  Read r writes its records to its
// own S slots (cnt[r] of them used); a read with more goes on in the overflow list.
template <bool CANON>
__global__ __launch_bounds__(SK_BLK) void sk_scan_kernel(ReadsView v, SkParams P, uint32_t max_len, SkRec* __restrict__ slots, uint32_t S,
                                                         uint8_t* __restrict__ cnt, SkRec* __restrict__ ovf, unsigned long long* __restrict__ ovf_cursor,
                                                         uint64_t ovf_cap) {
  extern __shared__ uint32_t sk_lds[];
  uint32_t* raw = sk_lds;                                                // [w][SK_BLK] order values of the current block
  uint32_t* suf = raw + P.w * SK_BLK;                                    // [w][SK_BLK] suffix minima of the block before
  const int tid = threadIdx.x;
  const int k = P.k, m = P.m, w = P.w;
  const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
  const uint64_t n_tiles = (v.n_reads + SK_BLK - 1) / SK_BLK;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t r = tile * SK_BLK + tid;
    if (r >= v.n_reads) continue;                                        // (no barrier in this kernel: the LDS columns are private)
    const uint32_t len = v.len ? v.len[r] : v.fixed_len;
    const uint64_t wb = v.woff ? v.woff[r] : r * v.wpr;
    const uint32_t nw = v.woff ? (uint32_t)(v.woff[r + 1] - v.woff[r]) : v.wpr;
    const uint64_t* rw = v.words + wb;
    SkRec* mine = slots + r * S;
    uint64_t cur = 0, curm = 0;
    uint32_t f = 0, rr = 0, pref = 0xFFFFFFFFu, run_ord = 0, nrec = 0;
    int lastN = -1, t = 0, p0 = 0;
    bool open = false;

    auto emit = [&](int a, int b, uint32_t ord) {                        // windows a..b of this read, minimizer order value ord
      SkRec rec;
      sk_extract96(rw, nw, (uint32_t)a, rec.hi, rec.lo);
      rec.meta = (uint32_t)(b - a + 1) | (sk_bucket(ord, P.bits) << 5);
      if (nrec < S) mine[nrec] = rec;
      else {
        const unsigned long long at = atomicAdd(ovf_cursor, 1ULL);
        if (at < ovf_cap) ovf[at] = rec;
      }
      nrec++;
    };

    const uint32_t stop = len < max_len ? len : max_len;
    for (uint32_t j = 0; j < stop; j++) {
      if ((j & 31) == 0) cur = rw[j >> 5];
      const uint32_t b = (uint32_t)(cur >> (62 - 2 * (j & 31))) & 3u;
      if (v.has_n) {
        if ((j & 63) == 0) curm = v.mask[wb / 2 + (j >> 6)];
        if ((curm >> (63 - (j & 63))) & 1ULL) lastN = (int)j;
      }
      f = ((f << 2) | b) & mmask;
      rr = (rr >> 2) | ((3u - b) << (2 * (m - 1)));
      if ((int)j < m - 1) continue;
      const int i = (int)j - m + 1;                                      // index of the m-mer that ends here; t = i mod w
      const uint32_t ord = sk_order(CANON ? (f < rr ? f : rr) : f);
      raw[t * SK_BLK + tid] = ord;
      pref = t == 0 ? ord : (ord < pref ? ord : pref);
      if (i >= w - 1) {
        uint32_t wmin = pref;
        if (t != w - 1) { const uint32_t sv = suf[(t + 1) * SK_BLK + tid]; wmin = sv < wmin ? sv : wmin; }
        const int p = i - w + 1;                                         // the window that is complete now
        const bool valid = lastN < p;
        if (open && (!valid || wmin != run_ord || p - p0 >= w)) { emit(p0, p - 1, run_ord); open = false; }
        if (valid && !open) { open = true; p0 = p; run_ord = wmin; }
      }
      if (t == w - 1) {                                                  // the block is complete: its suffix minima, for the next one
        uint32_t sm = 0xFFFFFFFFu;
        for (int tt = w - 1; tt >= 0; tt--) { const uint32_t x = raw[tt * SK_BLK + tid]; sm = x < sm ? x : sm; suf[tt * SK_BLK + tid] = sm; }
        t = 0;
      } else t++;
    }
    if (open) emit(p0, (int)len - k, run_ord);
    cnt[r] = (uint8_t)(nrec < S ? nrec : S);
  }
}

// ---------------------------------------------------------------- records to their buckets, one digit of the bucket id per pass
// Segment p of the input (segoff[p] .. segoff[p + 1]) is partitioned by digit = (bucket >> shift) & (nb - 1).  cnt != NULL: the
// input is the slot array of a read set -- item g is slot g % S of read g / S, used iff g % S < cnt[g / S]; cnt == NULL: a list of
// records, every entry with windows counts.
__device__ __forceinline__ bool skr_item(const SkRec* __restrict__ recs, const uint8_t* __restrict__ cnt, uint32_t S, uint64_t g, uint32_t& meta) {
  if (cnt) { const uint64_t r = g / S; if ((uint32_t)(g - r * S) >= cnt[r]) return false; }
  meta = recs[g].meta;
  return cnt || meta != 0;                              // (a list: an entry without windows is an unused one)
}
__global__ __launch_bounds__(SK_BLK) void skr_hist_kernel(const SkRec* __restrict__ recs, const uint8_t* __restrict__ cnt, uint32_t S,
                                                          const uint64_t* __restrict__ segoff, uint32_t n_seg, int shift, uint32_t nb,
                                                          uint32_t* __restrict__ hist) {
  extern __shared__ uint32_t skh[];
  for (uint32_t p = blockIdx.y; p < n_seg; p += gridDim.y) {
    const uint64_t s0 = segoff[p], s1 = segoff[p + 1];
    const uint64_t t0 = s0 + (uint64_t)blockIdx.x * SK_TILE;
    if (t0 >= s1) continue;
    const uint64_t t1 = min(t0 + (uint64_t)SK_TILE, s1);
    for (uint32_t i = threadIdx.x; i < nb; i += SK_BLK) skh[i] = 0;
    __syncthreads();
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += SK_BLK) {
      uint32_t meta;
      if (skr_item(recs, cnt, S, i, meta)) atomicAdd(&skh[((meta >> 5) >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += SK_BLK) if (skh[i]) atomicAdd(&hist[(uint64_t)p * nb + i], skh[i]);
    __syncthreads();
  }
}
// exclusive scan of each row of hist (one block per row) -> off (relative to the segment start) and a cursor copy
__global__ __launch_bounds__(SK_BLK) void skr_scan_rows_kernel(const uint32_t* __restrict__ hist, uint32_t nb2, uint32_t* __restrict__ off,
                                                               uint32_t* __restrict__ cursor) {
  __shared__ uint32_t part[SK_BLK];
  const uint64_t row = (uint64_t)blockIdx.x * nb2;
  const uint32_t per = (nb2 + SK_BLK - 1) / SK_BLK;
  uint32_t s = 0;
  for (uint32_t j = 0; j < per; j++) { const uint32_t i = threadIdx.x * per + j; if (i < nb2) s += hist[row + i]; }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t a = 0; for (int i = 0; i < SK_BLK; i++) { const uint32_t t = part[i]; part[i] = a; a += t; } }
  __syncthreads();
  uint32_t a = part[threadIdx.x];
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t i = threadIdx.x * per + j;
    if (i < nb2) { off[row + i] = a; cursor[row + i] = a; a += hist[row + i]; }
  }
}
// out_seg != NULL: the output of segment p starts at out_seg[p] (level 1: the slot arrays of several read sets into ONE record
// array); else at the segment's own start
__global__ __launch_bounds__(SK_BLK) void skr_scatter_kernel(const SkRec* __restrict__ recs, const uint8_t* __restrict__ cnt, uint32_t S,
                                                             const uint64_t* __restrict__ segoff, uint32_t n_seg, int shift, uint32_t nb,
                                                             uint32_t* __restrict__ cursor, const uint64_t* __restrict__ out_seg, SkRec* __restrict__ out) {
  extern __shared__ uint32_t sks[];
  uint32_t* lh = sks;
  uint32_t* lbase = sks + nb;
  for (uint32_t p = blockIdx.y; p < n_seg; p += gridDim.y) {
    const uint64_t s0 = segoff[p], s1 = segoff[p + 1];
    const uint64_t t0 = s0 + (uint64_t)blockIdx.x * SK_TILE;
    if (t0 >= s1) continue;
    const uint64_t t1 = min(t0 + (uint64_t)SK_TILE, s1);
    const uint64_t o0 = out_seg ? out_seg[p] : s0;
    for (uint32_t i = threadIdx.x; i < nb; i += SK_BLK) lh[i] = 0;
    __syncthreads();
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += SK_BLK) {
      uint32_t meta;
      if (skr_item(recs, cnt, S, i, meta)) atomicAdd(&lh[((meta >> 5) >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += SK_BLK) {
      const uint32_t c = lh[i];
      if (c) lbase[i] = atomicAdd(&cursor[(uint64_t)p * nb + i], c);
      lh[i] = 0;
    }
    __syncthreads();
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += SK_BLK) {
      uint32_t meta;
      if (!skr_item(recs, cnt, S, i, meta)) continue;
      const SkRec rec = recs[i];
      const uint32_t q = ((meta >> 5) >> shift) & (nb - 1);
      const uint32_t rank = atomicAdd(&lh[q], 1u);
      out[o0 + lbase[q] + rank] = rec;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- final buckets: windows -> keys -> LDS hash table -> pairs
// One block per bucket, eight lanes per record (lane q takes windows q, q + 8, ...).  Heavy keys would serialise on one LDS slot:
// two rounds of wave-level leader election fold the lanes holding the wave's first two distinct keys into one atomic each (as
// in buckets_kernel of count.hip).  The table is written to the pool -- a run reserved with one atomic -- at the end, and
// whenever it holds SK_SPILL distinct keys.
template <bool CANON>
__global__ __launch_bounds__(SK_BLK) void sk_buckets_kernel(const SkRec* __restrict__ recs, const uint64_t* __restrict__ off1,
                                                            const uint32_t* __restrict__ off2, const uint32_t* __restrict__ hist2, int b2, int k, int w,
                                                            uint64_t* __restrict__ pool_keys, uint32_t* __restrict__ pool_counts,
                                                            unsigned long long* __restrict__ pool_cursor, uint64_t pool_cap) {
  constexpr int FOLD_ROUNDS_ = 2;
  __shared__ unsigned long long tk[SK_CAP];
  __shared__ uint32_t tc[SK_CAP];
  __shared__ uint32_t lnew, lpos;
  __shared__ unsigned long long lbase;
  const uint32_t b = blockIdx.x;
  const uint32_t n = hist2[b];
  if (n == 0) return;
  const uint64_t s0 = off1[b >> b2] + off2[b];
  const int tid = threadIdx.x, lane = tid & 63, q = tid & 7, grp = tid >> 3;
  for (int i = tid; i < SK_CAP; i += SK_BLK) { tk[i] = SK_EMPTY; tc[i] = 0; }
  if (tid == 0) { lnew = 0; lpos = 0; }
  __syncthreads();
  const int iters = (w + 7) >> 3;

  auto spill = [&]() {                                                   // block-wide: table -> pool, table cleared
    const uint32_t nd = lnew;
    if (tid == 0) lbase = atomicAdd(pool_cursor, (unsigned long long)nd);
    __syncthreads();
    const unsigned long long base = lbase;
    const bool fits = base + nd <= pool_cap;
    for (int i = tid; i < SK_CAP; i += SK_BLK) {
      const unsigned long long key = tk[i];
      if (key != SK_EMPTY) {
        if (fits) { const uint32_t pos = atomicAdd(&lpos, 1u); pool_keys[base + pos] = key; pool_counts[base + pos] = tc[i]; }
        tk[i] = SK_EMPTY; tc[i] = 0;
      }
    }
    __syncthreads();
    if (tid == 0) { lnew = 0; lpos = 0; }
    __syncthreads();
  };

  for (uint32_t base = 0; base < n; base += SK_BLK / 8) {
    __syncthreads();
    const bool full = lnew > SK_SPILL;
    __syncthreads();
    if (full) spill();
    const uint32_t ri = base + grp;
    SkRec rec;
    rec.hi = 0; rec.lo = 0; rec.meta = 0;
    if (ri < n) rec = recs[s0 + ri];
    const int nwin = (int)(rec.meta & 31u);
    for (int it = 0; it < iters; it++) {
      const int j = q + 8 * it;
      bool active = j < nwin;
      unsigned long long key = 0;
      if (active) {
        key = sk_window(rec.hi, rec.lo, j, k);
        if (CANON) { const uint64_t rc = shn_revcomp(key, k); key = rc < key ? rc : key; }
      }
      uint32_t wgt = active ? 1u : 0u;
      bool elect = active;
#pragma unroll
      for (int round = 0; round < FOLD_ROUNDS_; round++) {
        const unsigned long long act = __ballot(elect);
        if (!act) break;
        const int leader = __ffsll((long long)act) - 1;
        const unsigned long long lk = __shfl(key, leader, 64);
        const bool same = elect && key == lk;
        const unsigned long long msk = __ballot(same);
        if (__popcll(msk) > 1) {
          if (lane == leader) wgt = (uint32_t)__popcll(msk); else if (same) active = false;
        }
        if (same) elect = false;
      }
      if (!active) continue;
      uint32_t slot = (uint32_t)(shn_mix64(key ^ 0x9E3779B97F4A7C15ULL)) & (SK_CAP - 1);
      for (int probe = 0; probe < SK_CAP; probe++) {
        const unsigned long long prev = atomicCAS(&tk[slot], SK_EMPTY, key);
        if (prev == SK_EMPTY) { atomicAdd(&lnew, 1u); atomicAdd(&tc[slot], wgt); break; }
        if (prev == key) { atomicAdd(&tc[slot], wgt); break; }
        slot = (slot + 1) & (SK_CAP - 1);
      }
    }
  }
  __syncthreads();
  if (lnew) spill();
}

// ---------------------------------------------------------------- final buckets, sorted: the table itself (layout 1)
// The same insertion, but the bucket's distinct keys leave the block SORTED, as a run of the table: buckets of minimizers ARE the
// buckets of the table (shn_table::layout 1), nothing is re-partitioned by key afterwards.  The table of a block holds CAP slots;
// a bucket with more distinct keys than that (a deeply covered locus drags its sequencing errors along: ~1.4 distinct k1-mers per
// read that covers it) is put on a list and done again by a block with a table of 8,192 slots, and what does not fit there is
// counted in passes over ranges of the key (the top 2, 4, ... bits): a pass holds complete counts of its range, and the passes'
// sorted runs follow each other in key order.  The runs are reserved in a pool with one atomic per bucket and copied into bucket
// order afterwards (sk_gather_kernel).
template <bool CANON, int THREADS, int CAP, int FOLD_ROUNDS_>
__global__ __launch_bounds__(THREADS) void sk_buckets_sorted_kernel(const SkRec* __restrict__ recs, const uint64_t* __restrict__ off1,
                                                                    const uint32_t* __restrict__ off2, const uint32_t* __restrict__ hist2, int b2, int k, int w,
                                                                    const uint32_t* __restrict__ list, uint32_t* __restrict__ defer, int defer_cursor,
                                                                    uint64_t* __restrict__ pool_keys, uint32_t* __restrict__ pool_counts,
                                                                    unsigned long long* __restrict__ cursors, uint64_t region_cap,
                                                                    unsigned long long* __restrict__ rcur,
                                                                    uint64_t* __restrict__ run_off, uint32_t* __restrict__ ndist) {
  extern __shared__ unsigned long long skb_lds[];
  unsigned long long* tk = skb_lds;                                      // [CAP]
  uint32_t* tc = (uint32_t*)(tk + CAP);                                  // [CAP]
  uint16_t* idx = (uint16_t*)(tc + CAP);                                 // [CAP]
  __shared__ uint32_t lnew, lover, lpos;
  __shared__ unsigned long long lbase;
  const uint32_t b = list ? list[blockIdx.x] : blockIdx.x;
  const uint32_t n = hist2[b];
  const int tid = threadIdx.x, lane = tid & 63, q = tid & 7, grp = tid >> 3;
  if (n == 0) { if (tid == 0) { ndist[b] = 0; run_off[b] = 0; } return; }
  const uint64_t s0 = off1[b >> b2] + off2[b];
  const int iters = (w + 7) >> 3;
  const uint32_t limit = (uint32_t)(CAP - (THREADS / 8) * w - 64);       // distinct keys a pass may reach before its last round

  // one pass over the bucket's records: the keys whose top `sbits` bits are `cls` into the (cleared) table; lover: it got too full
  auto pass = [&](uint32_t cls, int sbits) {
    for (int i = tid; i < CAP; i += THREADS) { tk[i] = SK_EMPTY; tc[i] = 0; }
    if (tid == 0) { lnew = 0; lover = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += THREADS / 8) {
      __syncthreads();
      const bool full = lnew > limit;
      __syncthreads();
      if (full) { if (tid == 0) lover = 1; break; }
      const uint32_t ri = base + grp;
      SkRec rec;
      rec.hi = 0; rec.lo = 0; rec.meta = 0;
      if (ri < n) rec = recs[s0 + ri];
      const int nwin = (int)(rec.meta & 31u);
      for (int it = 0; it < iters; it++) {
        const int j = q + 8 * it;
        bool active = j < nwin;
        unsigned long long key = 0;
        if (active) {
          key = sk_window(rec.hi, rec.lo, j, k);
          if (CANON) { const uint64_t rc = shn_revcomp(key, k); key = rc < key ? rc : key; }
          if (sbits && (uint32_t)(key >> (2 * k - sbits)) != cls) active = false;
        }
        uint32_t wgt = active ? 1u : 0u;
        bool elect = active;
#pragma unroll
        for (int round = 0; round < FOLD_ROUNDS_; round++) {
          const unsigned long long act = __ballot(elect);
          if (!act) break;
          const int leader = __ffsll((long long)act) - 1;
          const unsigned long long lk = __shfl(key, leader, 64);
          const bool same = elect && key == lk;
          const unsigned long long msk = __ballot(same);
          if (__popcll(msk) > 1) {
            if (lane == leader) wgt = (uint32_t)__popcll(msk); else if (same) active = false;
          }
          if (same) elect = false;
        }
        if (!active) continue;
        uint32_t slot = sk_slot_hash(key) & (CAP - 1);
        for (int probe = 0; probe < CAP; probe++) {
          const unsigned long long prev = atomicCAS(&tk[slot], SK_EMPTY, key);
          if (prev == SK_EMPTY) { atomicAdd(&lnew, 1u); atomicAdd(&tc[slot], wgt); break; }
          if (prev == key) { atomicAdd(&tc[slot], wgt); break; }
          slot = (slot + 1) & (CAP - 1);
        }
      }
    }
    __syncthreads();
  };
  // the table's keys, sorted, to pool[at ..)
  auto output = [&](uint64_t at) {
    const uint32_t nd = lnew;
    if (tid == 0) lpos = 0;
    __syncthreads();
    if (nd <= SK_RANK_MAX) {
      // few keys (the usual bucket holds ~170): every key counts the keys below it -- nd broadcast reads, no barrier -- and goes
      // straight to its place in the run; a bitonic sort of 256 keys is 36 barriers
      unsigned long long* dk = (unsigned long long*)idx;
      uint32_t* dc = (uint32_t*)(dk + SK_RANK_MAX);
      for (int i = tid; i < CAP; i += THREADS) {
        const unsigned long long key = tk[i];
        if (key != SK_EMPTY) { const uint32_t e = atomicAdd(&lpos, 1u); dk[e] = key; dc[e] = tc[i]; }
      }
      __syncthreads();
      for (uint32_t e = tid; e < nd; e += THREADS) {
        const unsigned long long key = dk[e];
        uint32_t r = 0;
        for (uint32_t j = 0; j < nd; j++) r += dk[j] < key ? 1u : 0u;
        pool_keys[at + r] = key; pool_counts[at + r] = dc[e];
      }
      __syncthreads();
      return;
    }
    for (int i = tid; i < CAP; i += THREADS) if (tk[i] != SK_EMPTY) idx[atomicAdd(&lpos, 1u)] = (uint16_t)i;
    uint32_t m2 = 1;
    while (m2 < nd) m2 <<= 1;
    __syncthreads();
    for (uint32_t i = nd + tid; i < m2; i += THREADS) idx[i] = 0xFFFFu;              // (beyond CAP: sorts last)
    __syncthreads();
    for (uint32_t kk = 2; kk <= m2; kk <<= 1) {
      for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
        for (uint32_t i = tid; i < m2; i += THREADS) {
          const uint32_t x = i ^ j;
          if (x > i) {
            const uint16_t ia = idx[i], ib = idx[x];
            const unsigned long long ka = ia == 0xFFFFu ? SK_EMPTY : tk[ia], kb = ib == 0xFFFFu ? SK_EMPTY : tk[ib];
            const bool asc = (i & kk) == 0;
            if ((ka > kb) == asc) { idx[i] = ib; idx[x] = ia; }
          }
        }
        __syncthreads();
      }
    }
    for (uint32_t i = tid; i < nd; i += THREADS) { const uint16_t ii = idx[i]; pool_keys[at + i] = tk[ii]; pool_counts[at + i] = tc[ii]; }
    __syncthreads();
  };

  int sbits = 0;
  uint32_t total = 0;
  for (;;) {                                                             // the number of range passes in which everything fits, and the distinct keys
    total = 0;
    bool ok = true;
    for (uint32_t cls = 0; cls < (1u << sbits); cls++) {
      pass(cls, sbits);
      if (lover) { ok = false; break; }
      total += lnew;
      if (sbits == 0) break;
      __syncthreads();
    }
    if (ok) break;
    if (defer) {                                                         // a block with a larger table does this bucket
      if (tid == 0) { const unsigned long long at = atomicAdd(&cursors[defer_cursor], 1ULL); defer[at] = b; ndist[b] = 0; run_off[b] = 0; }
      return;
    }
    sbits += 2;
    if (sbits > 16 || sbits > 2 * k) {                                   // (65,536 ranges and still too many keys in one: give up, the caller falls back)
      if (tid == 0) { atomicAdd(&cursors[3], 1ULL); ndist[b] = 0; run_off[b] = 0; }
      return;
    }
    __syncthreads();
  }
  // the run's place in the pool: the pool is SK_REGIONS stretches with a cursor each (bucket b uses stretch b mod SK_REGIONS).  One
  // cursor for all meant four million returning atomics on one address per launch -- they are served one after the other where the
  // address lives, and the kernel waited for them
  const uint32_t reg = b & (SK_REGIONS - 1);
  if (tid == 0) lbase = atomicAdd(&rcur[reg * SK_RSTRIDE], (unsigned long long)total);
  __syncthreads();
  const unsigned long long in_reg = lbase, base = (unsigned long long)reg * region_cap + in_reg;
  if (in_reg + total > region_cap) { if (tid == 0) { ndist[b] = 0; run_off[b] = 0; } return; }   // (the host sees the cursors and runs again with a larger pool)
  if (tid == 0) { ndist[b] = total; run_off[b] = base; }
  if (sbits == 0) { output(base); return; }
  uint64_t at = base;
  for (uint32_t cls = 0; cls < (1u << sbits); cls++) {
    pass(cls, sbits);
    const uint32_t nd = lnew;
    output(at);
    at += nd;
  }
}
// the buckets' runs from the pool into bucket order: one wavefront per bucket
__global__ __launch_bounds__(SK_BLK) void sk_gather_kernel(const uint64_t* __restrict__ pool_keys, const uint32_t* __restrict__ pool_counts,
                                                           const uint64_t* __restrict__ run_off, const uint32_t* __restrict__ ndist,
                                                           const uint64_t* __restrict__ boff, uint64_t n_buckets, uint64_t* __restrict__ keys,
                                                           uint32_t* __restrict__ counts) {
  const uint64_t b = ((uint64_t)blockIdx.x * SK_BLK + threadIdx.x) / SHN_WAVE;
  if (b >= n_buckets) return;
  const uint32_t lane = threadIdx.x & (SHN_WAVE - 1);
  const uint32_t nd = ndist[b];
  const uint64_t s0 = run_off[b], d0 = boff[b];
  for (uint32_t i = lane; i < nd; i += SHN_WAVE) { keys[d0 + i] = pool_keys[s0 + i]; counts[d0 + i] = pool_counts[s0 + i]; }
}
__global__ void sk_sum_counts_kernel(const uint32_t* __restrict__ c, uint64_t n, unsigned long long* __restrict__ out) {
  unsigned long long a = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a += c[i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0 && a) atomicAdd(out, a);
}

// ================================================================ host side
int shn_count_superkmers(shn_ctx* ctx, const std::vector<ReadsView>& views, uint64_t upper, int k1, int both_strands, shn_table** out, int* handled) {
  *handled = 0;
  const int mode = getenv("SHN_COUNT_SK") ? atoi(getenv("SHN_COUNT_SK")) : 1;          // 0 off, 1 large inputs, 2 always (tests)
  if (mode == 0 || k1 < SK_MIN_K || (k1 == 32 && !both_strands) || upper == 0) return SHN_OK;
  if (mode != 2 && upper < (1ULL << 22)) return SHN_OK;
  hipStream_t s = ctx->stream; shn_use_stream(s);
  SkParams P;
  P.k = k1;
  P.m = std::max(13, 2 * k1 - 48);
  P.w = k1 - P.m + 1;
  int bits = 6;
  while (bits < 22 && (upper >> bits) > 3500) bits++;
  if (getenv("SHN_COUNT_SK_BITS")) bits = std::max(4, std::min(22, atoi(getenv("SHN_COUNT_SK_BITS"))));      // (tests / experiments: any grid gives the same table)
  P.bits = bits;
  P.b1 = (bits + 1) / 2;
  P.b2 = bits - P.b1;
  const uint32_t nb1 = 1u << P.b1, nb2 = 1u << P.b2;
  const uint64_t nbk = 1ULL << bits;
  // the reads' slot arrays: S slots per read, a fifth more than the ~2 / (w + 1) records per window a read makes on average
  struct ViewPlan { uint64_t slot0, read0; uint32_t S, max_len; };
  std::vector<ViewPlan> plan(views.size());
  uint64_t n_slots = 0, n_reads = 0;
  for (size_t i = 0; i < views.size(); i++) {
    const ReadsView& v = views[i];
    const double expect = 2.0 * v.wmax / (P.w + 1) + 1.0;
    uint32_t S = ((uint32_t)(expect * 1.35 + 1.999) + 3u) & ~3u;
    S = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(S, 252), std::max<uint32_t>(v.wmax, 1)));
    if (getenv("SHN_COUNT_SK_SLOTS")) S = (uint32_t)std::max(1, std::min(252, atoi(getenv("SHN_COUNT_SK_SLOTS"))));     // (tests: reads that overflow their slots)
    plan[i] = ViewPlan{n_slots, n_reads, S, v.wmax ? v.wmax + (uint32_t)k1 - 1 : 0u};
    if (v.wmax && v.n_reads) { n_slots += v.n_reads * S; n_reads += v.n_reads; }
  }
  const uint64_t ovf_cap = n_reads / 2 + 4096;
  int rc;
  void *p, *pslots, *pcnt;
  // small arrays: level-1 histogram / offsets / cursors (u32), segment starts (u64), level-2 histogram / offsets / cursors, cursors of the
  // overflow list and of the pool
  const size_t small = (size_t)nb1 * 4 * 3 + 64 + ((size_t)nb1 + 4) * 8 + nbk * 4 * 3 + 64 + (views.size() + 2) * 16;
  if ((rc = shn_ws(ctx)[22].get(small, &p))) return rc;
  if ((rc = shn_ws(ctx)[18].get((n_slots + ovf_cap + 2) * sizeof(SkRec), &pslots))) return rc;
  if ((rc = shn_ws(ctx)[23].get(n_reads + 64, &pcnt))) return rc;
  uint32_t* d_hist1 = (uint32_t*)p;
  uint32_t* d_offr1 = d_hist1 + nb1;
  uint32_t* d_cursor1 = d_offr1 + nb1;
  unsigned long long* d_cursors = (unsigned long long*)(d_cursor1 + nb1 + ((nb1 & 1) ? 1 : 0));   // [0] overflow list, [1] pool, [4] = 0: where level 1's output starts
  uint64_t* d_seg = (uint64_t*)(d_cursors + 8);                          // [nb1 + 1] starts of the level-1 partitions
  uint32_t* d_hist2 = (uint32_t*)(d_seg + nb1 + 4);
  uint32_t* d_off2 = d_hist2 + nbk;
  uint32_t* d_cursor2 = d_off2 + nbk;
  uint64_t* d_pairs = (uint64_t*)(d_cursor2 + nbk + (nbk & 1));          // {0, items} per level-1 launch
  SkRec* slots = (SkRec*)pslots;
  SkRec* ovf = slots + n_slots;
  uint8_t* d_cnt = (uint8_t*)pcnt;
  HIP_TRY(hipMemsetAsync(d_hist1, 0, (size_t)nb1 * 4, s));
  HIP_TRY(hipMemsetAsync(d_cursors, 0, 64, s));
  HIP_TRY(hipMemsetAsync(ovf, 0, ovf_cap * sizeof(SkRec), s));            // (an unused overflow entry has meta 0 = no windows)
  const size_t lds_scan = (size_t)2 * P.w * SK_BLK * 4;
  for (size_t i = 0; i < views.size(); i++) {
    const ReadsView& v = views[i];
    if (!v.wmax || !v.n_reads) continue;
    TimerRegion t(ctx, T_SK_EMIT);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(v.n_reads, SK_BLK), 1u << 20);
    if (both_strands) hipLaunchKernelGGL(sk_scan_kernel<true>, dim3(grid), dim3(SK_BLK), lds_scan, s, v, P, plan[i].max_len, slots + plan[i].slot0, plan[i].S,
                                         d_cnt + plan[i].read0, ovf, d_cursors, ovf_cap);
    else hipLaunchKernelGGL(sk_scan_kernel<false>, dim3(grid), dim3(SK_BLK), lds_scan, s, v, P, plan[i].max_len, slots + plan[i].slot0, plan[i].S,
                            d_cnt + plan[i].read0, ovf, d_cursors, ovf_cap);
  }
  // level 1: the slot arrays (one launch per read set) and the overflow list into the partitions of ONE record array
  std::vector<uint64_t> pairs(2 * (views.size() + 1), 0);
  for (size_t i = 0; i <= views.size(); i++) pairs[2 * i + 1] = i == views.size() ? ovf_cap : ((views[i].wmax && views[i].n_reads) ? views[i].n_reads * plan[i].S : 0);
  HIP_TRY(hipMemcpyAsync(d_pairs, pairs.data(), pairs.size() * 8, hipMemcpyHostToDevice, s));
  auto level1 = [&](bool scatter, SkRec* recsA) {
    for (size_t i = 0; i <= views.size(); i++) {
      const bool is_ovf = i == views.size();
      const uint64_t n = pairs[2 * i + 1];
      if (!n) continue;
      const SkRec* in = is_ovf ? ovf : slots + plan[i].slot0;
      const uint8_t* c = is_ovf ? nullptr : d_cnt + plan[i].read0;
      const uint32_t S = is_ovf ? 1u : plan[i].S;
      const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, cdiv(n, SK_TILE));
      if (!scatter) hipLaunchKernelGGL(skr_hist_kernel, dim3(tiles, 1), dim3(SK_BLK), nb1 * 4, s, in, c, S, d_pairs + 2 * i, 1u, P.b2, nb1, d_hist1);
      else hipLaunchKernelGGL(skr_scatter_kernel, dim3(tiles, 1), dim3(SK_BLK), nb1 * 8, s, in, c, S, d_pairs + 2 * i, 1u, P.b2, nb1, d_cursor1,
                              (const uint64_t*)(d_cursors + 4), recsA);
    }
  };
  {
    TimerRegion t(ctx, T_SK_HIST);
    level1(false, nullptr);
  }
  std::vector<uint32_t> h1(nb1);
  unsigned long long ovf_used = 0;
  HIP_TRY(hipMemcpyAsync(h1.data(), d_hist1, (size_t)nb1 * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&ovf_used, d_cursors, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (ovf_used > ovf_cap) return SHN_OK;                                 // (reads that make far more records than their windows let expect: the partition pipeline takes them)
  std::vector<uint64_t> off1(nb1 + 1);
  std::vector<uint32_t> off1r(nb1);
  uint64_t NR = 0, max1 = 0;
  for (uint32_t i = 0; i < nb1; i++) { off1[i] = NR; NR += h1[i]; max1 = std::max<uint64_t>(max1, h1[i]); }
  off1[nb1] = NR;
  if (NR >= 0xFFFFFFFFULL) return SHN_OK;                                // (2^32 records: leave it to the chunked pipeline)
  for (uint32_t i = 0; i < nb1; i++) off1r[i] = (uint32_t)off1[i];
  void* pa;
  if ((rc = shn_ws(ctx)[19].get((NR + 2) * sizeof(SkRec), &pa))) return rc;
  SkRec* recsA = (SkRec*)pa;
  SkRec* recsB = slots;                                                  // (level 2 writes where the slots were: n_slots + overflow >= NR)
  HIP_TRY(hipMemcpyAsync(d_cursor1, off1r.data(), (size_t)nb1 * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_seg, off1.data(), (size_t)(nb1 + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_hist2, 0, nbk * 4, s));
  HIP_TRY(hipStreamSynchronize(s));
  {
    TimerRegion t(ctx, T_SK_HIST);
    level1(true, recsA);
  }
  const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, cdiv(max1, SK_TILE));
  {
    TimerRegion t(ctx, T_SK_HIST2);
    hipLaunchKernelGGL(skr_hist_kernel, dim3(tiles, nb1), dim3(SK_BLK), nb2 * 4, s, recsA, (const uint8_t*)nullptr, 1u, d_seg, nb1, 0, nb2, d_hist2);
  }
  hipLaunchKernelGGL(skr_scan_rows_kernel, dim3(nb1), dim3(SK_BLK), 0, s, d_hist2, nb2, d_off2, d_cursor2);
  {
    TimerRegion t(ctx, T_SK_SCATTER2);
    hipLaunchKernelGGL(skr_scatter_kernel, dim3(tiles, nb1), dim3(SK_BLK), nb2 * 8, s, recsA, (const uint8_t*)nullptr, 1u, d_seg, nb1, 0, nb2, d_cursor2,
                       (const uint64_t*)nullptr, recsB);
  }
  // the pool of (key, count) pairs: sized from what the buckets emitted per window last time on this context (an eighth of the
  // windows to begin with); a pool that turns out too small is made as large as the cursor says and the bucket kernels run again
  double ratio = ctx->sk_pool_ratio > 0 ? ctx->sk_pool_ratio * 1.05 : 0.125;
  if (getenv("SHN_COUNT_SK_POOL")) ratio = atof(getenv("SHN_COUNT_SK_POOL"));                                  // (tests: start too small)
  uint64_t cap = std::min<uint64_t>(upper, (uint64_t)((double)upper * ratio) + 4096);
  uint64_t np = 0;
  void *pk = nullptr, *pc = nullptr;
  // layout 1 (default): the buckets' sorted runs ARE the table; SHN_COUNT_SK_LAYOUT=0 (and the fallback of a bucket that 65,536 key
  // ranges do not split): unsorted pairs, reduced and re-partitioned by key through the pairs path
  bool sorted = !(getenv("SHN_COUNT_SK_LAYOUT") && atoi(getenv("SHN_COUNT_SK_LAYOUT")) == 0);
  if (sorted) {
    void* pr;
    if ((rc = shn_ws(ctx)[24].get(nbk * 20 + 64, &pr))) return rc;
    uint64_t* d_run_off = (uint64_t*)pr;
    uint32_t* d_ndist = (uint32_t*)(d_run_off + nbk);
    uint32_t* d_defer = d_ndist + nbk;
    uint32_t* d_defer2 = d_defer + nbk;
    // Three sizes of table.  A block's time is latency (the bucket's offsets, its records, the pool cursor: dependent round trips)
    // hidden by the blocks beside it, and the LDS a table takes sets how many are beside it: 1,024 slots (18 KB, eight blocks per
    // CU) do the 93 % of the buckets with at most 512 distinct keys in half the time 2,048 slots (30 KB, five blocks) took for all
    // of them; the rest go to 2,048 slots, what does not fit there (a deeply covered locus) to blocks of 1,024 threads and 8,192
    // slots, which fold equal keys of a wavefront before they reach the table.  (measured: tools/count_time.py)
    constexpr int T1_CAP = SK_T1CAP, T2_CAP = SK_SCAP, BIG_T = 1024, BIG_CAP = 8192;
    const size_t lds_t1 = (size_t)T1_CAP * 12 + std::max<size_t>((size_t)T1_CAP * 2, (size_t)SK_RANK_MAX * 12),
                 lds_t2 = (size_t)T2_CAP * 12 + std::max<size_t>((size_t)T2_CAP * 2, (size_t)SK_RANK_MAX * 12), lds_big = (size_t)BIG_CAP * 14;
    HIP_TRY(hipFuncSetAttribute((const void*)sk_buckets_sorted_kernel<true, BIG_T, BIG_CAP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
    HIP_TRY(hipFuncSetAttribute((const void*)sk_buckets_sorted_kernel<false, BIG_T, BIG_CAP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
    unsigned long long cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void* prc;
    if ((rc = shn_ws(ctx)[29].get((size_t)SK_REGIONS * SK_RSTRIDE * 8, &prc))) return rc;
    unsigned long long* d_rcur = (unsigned long long*)prc;
    std::vector<unsigned long long> h_rcur((size_t)SK_REGIONS * SK_RSTRIDE);
    for (int attempt = 0; attempt < 3 && sorted; attempt++) {
      const uint64_t region_cap = cap / SK_REGIONS + 64;
      if ((rc = shn_ws(ctx)[20].get((region_cap * SK_REGIONS + 2) * 8, &pk))) return rc;
      if ((rc = shn_ws(ctx)[21].get((region_cap * SK_REGIONS + 2) * 4, &pc))) return rc;
      HIP_TRY(hipMemsetAsync(d_cursors + 1, 0, 56, s));                  // [1] unused, [2] / [6] buckets put off to the next size, [3] given up, [5] sum of counts
      HIP_TRY(hipMemsetAsync(d_rcur, 0, (size_t)SK_REGIONS * SK_RSTRIDE * 8, s));
      {
#define SK_LAUNCH(CANON_, T_, CAP_, FOLD_, GRID_, LDS_, LIST_, DEFER_, DCUR_) \
        hipLaunchKernelGGL((sk_buckets_sorted_kernel<CANON_, T_, CAP_, FOLD_>), dim3((uint32_t)(GRID_)), dim3(T_), LDS_, s, recsB, d_seg, d_off2, d_hist2, P.b2, k1, P.w, \
                           (const uint32_t*)(LIST_), (uint32_t*)(DEFER_), DCUR_, (uint64_t*)pk, (uint32_t*)pc, d_cursors, region_cap, d_rcur, d_run_off, d_ndist)
        { TimerRegion t(ctx, T_SK_BUCKETS);                              // (one region per table size: three kernels of very different cost)
          if (both_strands) SK_LAUNCH(true, SK_STHREADS, T1_CAP, 0, nbk, lds_t1, nullptr, d_defer, 2);
          else SK_LAUNCH(false, SK_STHREADS, T1_CAP, 0, nbk, lds_t1, nullptr, d_defer, 2); }
        HIP_TRY(hipMemcpyAsync(cur, d_cursors, 64, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const unsigned long long n_t2 = cur[2];
        if (n_t2) {
          TimerRegion t2(ctx, T_SK_BUCKETS2);
          if (both_strands) SK_LAUNCH(true, SK_T2THREADS, T2_CAP, 0, n_t2, lds_t2, d_defer, d_defer2, 6);
          else SK_LAUNCH(false, SK_T2THREADS, T2_CAP, 0, n_t2, lds_t2, d_defer, d_defer2, 6);
          HIP_TRY(hipMemcpyAsync(cur, d_cursors, 64, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
        }
        if (getenv("SHN_COUNT_SK_DEBUG")) fprintf(stderr, "[sk] buckets %llu, %llu of them to 2,048 slots, %llu to 8,192; pool of %llu pairs in %d stretches\n", (unsigned long long)nbk, n_t2, cur[6], (unsigned long long)cap, SK_REGIONS);
        if (cur[6]) {
          TimerRegion tb(ctx, T_SK_BIG);
          if (both_strands) SK_LAUNCH(true, BIG_T, BIG_CAP, 2, cur[6], lds_big, d_defer2, nullptr, 2);
          else SK_LAUNCH(false, BIG_T, BIG_CAP, 2, cur[6], lds_big, d_defer2, nullptr, 2);
          HIP_TRY(hipMemcpyAsync(cur, d_cursors, 64, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
        }
#undef SK_LAUNCH
      }
      HIP_TRY(hipGetLastError());
      if (cur[3]) { sorted = false; break; }                             // a bucket no key range splits: the pairs path takes this input
      HIP_TRY(hipMemcpyAsync(h_rcur.data(), d_rcur, h_rcur.size() * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      unsigned long long fullest = 0;
      np = 0;
      for (int r_ = 0; r_ < SK_REGIONS; r_++) { const unsigned long long u = h_rcur[(size_t)r_ * SK_RSTRIDE]; np += u; fullest = std::max(fullest, u); }
      if (fullest <= region_cap) break;
      if (attempt == 2) return shn_fail(SHN_ERR_OVERFLOW, "shn_count_k1mers: the pair pool of the super-k-mer path stayed too small");
      cap = std::max<uint64_t>(np + np / 64 + 4096, (fullest + fullest / 64 + 64) * SK_REGIONS);          // (every stretch as large as the fullest asked for)
    }
    if (sorted) {
      if (!getenv("SHN_COUNT_SK_POOL")) ctx->sk_pool_ratio = (double)np / (double)upper;
      shn_table* t = new shn_table();
      memset(t, 0, sizeof(*t));
      t->ctx = ctx; t->device = ctx->device; t->k = k1; t->canonical = both_strands ? 1 : 0; t->bits = bits; t->n_buckets = nbk;
      t->layout = 1; t->sk_m = P.m; t->n = np;
#define TRYT(x) do { hipError_t _e = (x); if (_e != hipSuccess) { shn_table_destroy(t); return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
      TRYT(shn_dev_malloc(&t->d_keys, (np + 1) * 8));
      TRYT(shn_dev_malloc(&t->d_counts, (np + 1) * 4));
      TRYT(shn_dev_malloc(&t->d_bucket_off, (nbk + 1) * 8));
      uint64_t D = 0;
      { TimerRegion tr(ctx, T_COMPACT);
        if ((rc = shn_device_scan_u32(ctx, d_ndist, nbk, t->d_bucket_off, &D))) { shn_table_destroy(t); return rc; }
        if (D != np) { shn_table_destroy(t); return shn_fail(SHN_ERR_OVERFLOW, "shn_count_k1mers: super-k-mer buckets and pool disagree"); }
        if (np) hipLaunchKernelGGL(sk_gather_kernel, dim3((uint32_t)cdiv(nbk * SHN_WAVE, SK_BLK)), dim3(SK_BLK), 0, s, (const uint64_t*)pk, (const uint32_t*)pc,
                                   (const uint64_t*)d_run_off, (const uint32_t*)d_ndist, (const uint64_t*)t->d_bucket_off, nbk, t->d_keys, t->d_counts);
        TRYT(hipMemsetAsync(d_cursors + 5, 0, 8, s));
        if (np) hipLaunchKernelGGL(sk_sum_counts_kernel, dim3(1024), dim3(256), 0, s, (const uint32_t*)t->d_counts, np, d_cursors + 5);   // (the pool has gaps between its stretches: the table's counts)
        unsigned long long tot = 0;
        TRYT(hipMemcpyAsync(&tot, d_cursors + 5, 8, hipMemcpyDeviceToHost, s));
        TRYT(hipStreamSynchronize(s));
        TRYT(hipGetLastError());
        t->total = tot;
      }
#undef TRYT
      *out = t;
      *handled = 1;
      return SHN_OK;
    }
  }
  for (int attempt = 0; attempt < 3; attempt++) {
    if ((rc = shn_ws(ctx)[20].get((cap + 2) * 8, &pk))) return rc;
    if ((rc = shn_ws(ctx)[21].get((cap + 2) * 4, &pc))) return rc;
    HIP_TRY(hipMemsetAsync(d_cursors + 1, 0, 8, s));
    {
      TimerRegion t(ctx, T_SK_BUCKETS);
      if (both_strands) hipLaunchKernelGGL(sk_buckets_kernel<true>, dim3((uint32_t)nbk), dim3(SK_BLK), 0, s, recsB, d_seg, d_off2, d_hist2, P.b2, k1, P.w,
                                           (uint64_t*)pk, (uint32_t*)pc, d_cursors + 1, cap);
      else hipLaunchKernelGGL(sk_buckets_kernel<false>, dim3((uint32_t)nbk), dim3(SK_BLK), 0, s, recsB, d_seg, d_off2, d_hist2, P.b2, k1, P.w,
                              (uint64_t*)pk, (uint32_t*)pc, d_cursors + 1, cap);
    }
    unsigned long long used = 0;
    HIP_TRY(hipMemcpyAsync(&used, d_cursors + 1, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    np = used;
    if (np <= cap) break;
    if (attempt == 2) return shn_fail(SHN_ERR_OVERFLOW, "shn_count_k1mers: the pair pool of the super-k-mer path stayed too small");
    cap = std::min<uint64_t>(upper, np + np / 64 + 4096);
  }
  if (np >= 0xFFFFFFFFULL) return SHN_OK;                                // (more pairs than the pairs path takes: the chunked pipeline)
  rc = shn_table_from_pairs(ctx, pk, pc, np, k1, both_strands ? 1 : 0, out);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  *handled = 1;
  return SHN_OK;
}
