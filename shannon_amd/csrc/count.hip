// (K+1)-mer counting on gfx950: replaces `jellyfish count | dump` (shannon.py:439-441).
//
// Pipeline (all HBM-bound byte/integer work, no MFMA):
//   hist1     packed reads -> canonical keys (recomputed, never stored) -> level-1 bucket histogram
//   scatter1  packed reads -> keys -> level-1 partitions   (keysA)      [8 B written per window]
//   hist2     keysA        -> per-partition level-2 histogram
//   scatter2  keysA        -> final buckets               (keysB)
//   buckets   keysB        -> per-bucket LDS hash-aggregate + LDS bitonic sort -> (key,count) runs
//   compact   runs         -> dense table (grouped by bucket, ascending inside a bucket)
// Buckets are ranges of the top `bits` bits of murmur-fmix64(key); one bucket fits an LDS hash
// table of CAP slots, so duplicates (heavy k-mers) cost no capacity, only LDS atomics.
#include "common.h"
#include <mutex>
#include <atomic>
#include <thread>
#include <functional>
#include <cmath>
#include <algorithm>
#include <cstring>

#define BLK 256
#define CAP 1024            // LDS hash slots per final bucket (2048: 48 KB of LDS per block = 3 blocks per CU; 1024: 6)
#define CAP_LIMIT 950       // distinct keys per bucket before we call it overflow
#define TARGET_BUCKET 640   // average windows per final bucket
#define TILE_IDS 65536      // window ids per block tile in hist1/scatter1
#define TILE_KEYS 32768     // keys per block tile in hist2/scatter2
#define EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL

#include "count_views.h"

#define bucket_of shn_bucket_of

template <bool CANON>
__device__ __forceinline__ bool gen_key(const ReadsView& v, uint64_t r, uint32_t pos, int k, uint64_t& key) {
  uint32_t len = v.len ? v.len[r] : v.fixed_len;
  if (pos + k > len) return false;
  uint64_t wb = v.woff ? v.woff[r] : r * v.wpr;
  if (v.has_n && shn_extract_mask(v.mask + wb / 2, pos, k)) return false;
  key = shn_extract(v.words + wb, pos, k);
  if (CANON) { uint64_t rc = shn_revcomp(key, k); key = rc < key ? rc : key; }
  return true;
}


// ---------------------------------------------------------------- level 1 (from packed reads)
// Besides the level-1 histogram the pass fills a HyperLogLog sketch (2^12 registers) of the keys: the number of
// DISTINCT k-mers decides how many final buckets are needed (a bucket's LDS table holds distinct keys), and at
// high coverage that is orders of magnitude below the number of windows.
#define HLL_BITS 12
template <bool CANON>
__global__ __launch_bounds__(BLK) void hist1_kernel(ReadsView v, int k, int bits, int b2, uint64_t n_tiles,
                                                    unsigned long long* __restrict__ hist1, uint32_t* __restrict__ hll) {
  extern __shared__ uint32_t lh[];
  __shared__ uint32_t lreg[1 << HLL_BITS];
  const int nb1 = 1 << (bits - b2);
  for (int i = threadIdx.x; i < nb1; i += BLK) lh[i] = 0;
  for (int i = threadIdx.x; i < (1 << HLL_BITS); i += BLK) lreg[i] = 0;
  __syncthreads();
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint64_t r0 = tile * v.rt;
    uint32_t nr = (uint32_t)min((uint64_t)v.rt, v.n_reads - r0);
    uint32_t nid = nr * v.wmax;
    for (uint32_t i = threadIdx.x; i < nid; i += BLK) {
      uint32_t rl = i / v.wmax, pos = i - rl * v.wmax;
      uint64_t key;
      if (gen_key<CANON>(v, r0 + rl, pos, k, key)) {
        const uint64_t h = shn_mix64(key);
        atomicAdd(&lh[bits ? (uint32_t)(h >> (64 - bits)) >> b2 : 0u], 1u);
        // register = low 12 bits, rank = leading zeros of the next 40 bits + 1 (the bucket uses the top bits)
        const uint32_t reg = (uint32_t)h & ((1u << HLL_BITS) - 1);
        const uint64_t rest = (h >> HLL_BITS) & ((1ULL << 40) - 1);
        const uint32_t rho = rest ? (uint32_t)(__clzll((long long)rest) - 24 + 1) : 41u;
        if (lreg[reg] < rho) atomicMax(&lreg[reg], rho);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb1; i += BLK)
    if (lh[i]) atomicAdd(&hist1[i], (unsigned long long)lh[i]);
  for (int i = threadIdx.x; i < (1 << HLL_BITS); i += BLK)
    if (lreg[i]) atomicMax(&hll[i], lreg[i]);
}

template <bool CANON>
__global__ __launch_bounds__(BLK) void scatter1_kernel(ReadsView v, int k, int bits, int b2, uint64_t n_tiles,
                                                       unsigned long long* __restrict__ cursor1,
                                                       uint64_t* __restrict__ out) {
  extern __shared__ uint32_t lds[];
  const int nb1 = 1 << (bits - b2);
  unsigned long long* lbase = (unsigned long long*)lds;          // [nb1] reserved global bases
  uint32_t* lh = (uint32_t*)(lbase + nb1);                        // [nb1] counts, then running ranks
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    for (int i = threadIdx.x; i < nb1; i += BLK) lh[i] = 0;
    __syncthreads();
    uint64_t r0 = tile * v.rt;
    uint32_t nr = (uint32_t)min((uint64_t)v.rt, v.n_reads - r0);
    uint32_t nid = nr * v.wmax;
    // four windows per thread and trip: the key generation, the LDS atomics and the stores of the four are independent,
    // so their latencies overlap (the loop is latency-bound at full occupancy)
    for (uint32_t i0 = threadIdx.x; i0 < nid; i0 += 4 * BLK) {
      uint64_t key[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * BLK;
        const uint32_t rl = i / v.wmax, pos = i - rl * v.wmax;
        ok[u] = i < nid && gen_key<CANON>(v, r0 + rl, pos, k, key[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) if (ok[u]) atomicAdd(&lh[bucket_of(key[u], bits) >> b2], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb1; i += BLK) {
      uint32_t c = lh[i];
      if (c) lbase[i] = atomicAdd(&cursor1[i], (unsigned long long)c);
      lh[i] = 0;
    }
    __syncthreads();
    for (uint32_t i0 = threadIdx.x; i0 < nid; i0 += 4 * BLK) {
      uint64_t key[4];
      uint32_t p[4], rank[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * BLK;
        const uint32_t rl = i / v.wmax, pos = i - rl * v.wmax;
        ok[u] = i < nid && gen_key<CANON>(v, r0 + rl, pos, k, key[u]);
        p[u] = ok[u] ? bucket_of(key[u], bits) >> b2 : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) if (ok[u]) rank[u] = atomicAdd(&lh[p[u]], 1u);
#pragma unroll
      for (int u = 0; u < 4; u++) if (ok[u]) out[lbase[p[u]] + rank[u]] = key[u];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- levels from a key array
// A level partitions every SEGMENT of the key array (segoff[p] .. segoff[p+1]) by one digit of the keys' bucket id:
// digit = (bucket >> shift) & (nb - 1).  No level fans out into more than 256 streams: a block then writes runs of >= 1 KB per
// stream and tile, which the L2 merges into whole lines -- with 2,048-4,096 streams (two levels of 11-12 bits at 2^23 buckets) a
// run was 16 keys, the lines left the L2 half written and the scatter kernels wrote 3-3.6 times their bytes.
__device__ __forceinline__ uint32_t digit_at(uint64_t key, int bits, int shift, uint32_t nb) {
  return (bucket_of(key, bits) >> shift) & (nb - 1);
}

__global__ __launch_bounds__(BLK) void hist_keys_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ segoff, uint32_t n_seg,
                                                        int bits, int shift, uint32_t nb, uint32_t* __restrict__ hist, uint32_t tile_keys) {
  extern __shared__ uint32_t lh[];
  for (uint32_t p = blockIdx.y; p < n_seg; p += gridDim.y) {
    const uint64_t s0 = segoff[p], s1 = segoff[p + 1];
    uint64_t t0 = s0 + (uint64_t)blockIdx.x * tile_keys;
    if (t0 >= s1) continue;                                  // (uniform over the block)
    uint64_t t1 = min(t0 + (uint64_t)tile_keys, s1);
    for (uint32_t i = threadIdx.x; i < nb; i += BLK) lh[i] = 0;
    __syncthreads();
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += BLK)
      atomicAdd(&lh[digit_at(keys[i], bits, shift, nb)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += BLK)
      if (lh[i]) atomicAdd(&hist[(uint64_t)p * nb + i], lh[i]);
    __syncthreads();
  }
}

// exclusive scan of each row of hist -> off (relative to the segment start) and cursor copy
__global__ __launch_bounds__(BLK) void scan_rows_kernel(const uint32_t* __restrict__ hist2, int nb2,
                                                        uint32_t* __restrict__ off2, uint32_t* __restrict__ cursor2) {
  __shared__ uint32_t part[BLK];
  const uint64_t row = (uint64_t)blockIdx.x * nb2;
  const int per = (nb2 + BLK - 1) / BLK;
  uint32_t s = 0;
  for (int j = 0; j < per; j++) { int i = threadIdx.x * per + j; if (i < nb2) s += hist2[row + i]; }
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t a = 0; for (int i = 0; i < BLK; i++) { uint32_t t = part[i]; part[i] = a; a += t; } }
  __syncthreads();
  uint32_t a = part[threadIdx.x];
  for (int j = 0; j < per; j++) {
    int i = threadIdx.x * per + j;
    if (i < nb2) { off2[row + i] = a; cursor2[row + i] = a; a += hist2[row + i]; }
  }
}

template <bool HASC>
__global__ __launch_bounds__(BLK) void scatter_keys_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ cin,
                                                           const uint64_t* __restrict__ segoff, uint32_t n_seg, int bits, int shift, uint32_t nb,
                                                           uint32_t* __restrict__ cursor, uint64_t* __restrict__ out,
                                                           uint32_t* __restrict__ cout, uint32_t tile_keys) {
  extern __shared__ uint32_t lds[];
  uint32_t* lh = lds;
  uint32_t* lbase = lds + nb;
  for (uint32_t p = blockIdx.y; p < n_seg; p += gridDim.y) {
    const uint64_t s0 = segoff[p], s1 = segoff[p + 1];
    uint64_t t0 = s0 + (uint64_t)blockIdx.x * tile_keys;
    if (t0 >= s1) continue;
    uint64_t t1 = min(t0 + (uint64_t)tile_keys, s1);
    for (uint32_t i = threadIdx.x; i < nb; i += BLK) lh[i] = 0;
    __syncthreads();
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += BLK)
      atomicAdd(&lh[digit_at(keys[i], bits, shift, nb)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += BLK) {
      uint32_t c = lh[i];
      if (c) lbase[i] = atomicAdd(&cursor[(uint64_t)p * nb + i], c);
      lh[i] = 0;
    }
    __syncthreads();
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += BLK) {
      uint64_t key = keys[i];
      uint32_t q = digit_at(key, bits, shift, nb);
      uint32_t rank = atomicAdd(&lh[q], 1u);
      out[s0 + lbase[q] + rank] = key;
      if (HASC) cout[s0 + lbase[q] + rank] = cin[i];
    }
    __syncthreads();
  }
}

// absolute start of every segment of the NEXT level: the segments of a level are the (segment, digit) pairs of the one before
__global__ void next_segments_kernel(const uint64_t* __restrict__ segoff, uint32_t n_seg, const uint32_t* __restrict__ off_rel, uint32_t nb,
                                     uint64_t* __restrict__ out) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t n = (uint64_t)n_seg * nb;
  if (q < n) out[q] = segoff[q / nb] + off_rel[q];
  else if (q == n) out[q] = segoff[n_seg];
}

// ---------------------------------------------------------------- final buckets
// One block per final bucket: LDS hash-aggregate, compact, bitonic sort, write runs in place.
template <bool HASC>
__global__ __launch_bounds__(BLK) void buckets_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ cin,
                                                      const uint64_t* __restrict__ off1,
                                                      const uint32_t* __restrict__ off2, const uint32_t* __restrict__ hist2,
                                                      int b2, uint64_t* __restrict__ out_keys, uint32_t* __restrict__ out_counts,
                                                      uint32_t* __restrict__ ndist, uint32_t* __restrict__ overflow) {
  __shared__ unsigned long long tk[CAP];
  __shared__ uint32_t tc[CAP];
  __shared__ unsigned long long ck[CAP];
  __shared__ uint32_t cc[CAP];
  __shared__ uint32_t lcount, lall;
  const uint32_t b = blockIdx.x;
  const uint32_t p = b >> b2;
  const uint32_t n = hist2[b];
  if (n == 0) { if (threadIdx.x == 0) ndist[b] = 0; return; }
  const uint64_t s0 = off1[p] + off2[b];
  for (int i = threadIdx.x; i < CAP; i += BLK) { tk[i] = EMPTY_KEY; tc[i] = 0; }
  if (threadIdx.x == 0) { lcount = 0; lall = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < n; base += BLK) {
    const uint32_t i = base + threadIdx.x;
    bool active = i < n;
    unsigned long long key = active ? keys[s0 + i] : 0ULL;
    uint32_t w = active ? (HASC ? cin[s0 + i] : 1u) : 0u;
    if (active && key == EMPTY_KEY) { atomicAdd(&lall, w); active = false; }      // k=32, non-canonical all-T
    // heavy k-mers (77,000x coverage in config[1]) would serialise on one LDS slot: two rounds of wave-level
    // leader election fold the lanes holding the wave's first two distinct keys into one atomic each
    const int lane = threadIdx.x & 63;
    bool elect = active;                       // lanes whose key has not been looked at yet
#pragma unroll
    for (int round = 0; round < 2; round++) {
      unsigned long long act = __ballot(elect);
      if (!act) break;
      int leader = __ffsll((long long)act) - 1;
      unsigned long long lk = __shfl(key, leader, 64);
      bool same = elect && key == lk;
      unsigned long long m = __ballot(same);
      if (__popcll(m) > 1) {
        uint32_t tot = same ? w : 0u;
        if (HASC) { for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64); }
        else tot = (uint32_t)__popcll(m);
        if (lane == leader) w = tot; else if (same) active = false;
      }
      if (same) elect = false;
    }
    if (!active) continue;
    uint32_t slot = (uint32_t)(shn_mix64(key ^ 0x9E3779B97F4A7C15ULL)) & (CAP - 1);
    for (int probe = 0; probe < CAP; probe++) {
      unsigned long long prev = atomicCAS(&tk[slot], EMPTY_KEY, key);
      if (prev == EMPTY_KEY || prev == key) { atomicAdd(&tc[slot], w); break; }
      slot = (slot + 1) & (CAP - 1);
      if (probe == CAP - 1) atomicExch(overflow, 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CAP; i += BLK) {
    if (tk[i] != EMPTY_KEY) { uint32_t pos = atomicAdd(&lcount, 1u); ck[pos] = tk[i]; cc[pos] = tc[i]; }
  }
  __syncthreads();
  uint32_t nd = lcount;
  if (nd > CAP_LIMIT) { if (threadIdx.x == 0) atomicExch(overflow, 1u); }
  uint32_t m = 1;
  while (m < nd) m <<= 1;
  for (uint32_t i = nd + threadIdx.x; i < m; i += BLK) { ck[i] = EMPTY_KEY; cc[i] = 0; }
  __syncthreads();
  for (uint32_t kk = 2; kk <= m; kk <<= 1) {
    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < m; i += BLK) {
        uint32_t x = i ^ j;
        if (x > i) {
          bool asc = (i & kk) == 0;
          unsigned long long a = ck[i], c = ck[x];
          if ((a > c) == asc) { ck[i] = c; ck[x] = a; uint32_t t = cc[i]; cc[i] = cc[x]; cc[x] = t; }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = threadIdx.x; i < nd; i += BLK) { out_keys[s0 + i] = ck[i]; out_counts[s0 + i] = cc[i]; }
  if (lall) {   // the all-ones key sorts last
    if (threadIdx.x == 0) { out_keys[s0 + nd] = EMPTY_KEY; out_counts[s0 + nd] = lall; }
    nd += 1;
  }
  if (threadIdx.x == 0) ndist[b] = nd;
}

// ---------------------------------------------------------------- generic exclusive scan u32 -> u64
__global__ __launch_bounds__(BLK) void scan_block_sums(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ sums) {
  __shared__ unsigned long long red[BLK];
  uint64_t base = (uint64_t)blockIdx.x * 1024;
  unsigned long long s = 0;
  for (int j = 0; j < 4; j++) { uint64_t i = base + threadIdx.x * 4 + j; if (i < n) s += in[i]; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = BLK / 2; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
// exclusive scan of the block sums in place, one block: every thread scans a contiguous chunk, the chunk totals are
// scanned in LDS (a single thread walking the whole array was 74 us per call, and the radix sort calls it per pass)
__global__ __launch_bounds__(1024) void scan_sums_serial(uint64_t* sums, uint64_t nblocks, uint64_t* total) {
  __shared__ unsigned long long part[1024];
  const uint64_t per = (nblocks + 1023) / 1024;
  const uint64_t lo = min(nblocks, (uint64_t)threadIdx.x * per), hi = min(nblocks, lo + per);
  unsigned long long s = 0;
  for (uint64_t i = lo; i < hi; i++) s += sums[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int st = 1; st < 1024; st <<= 1) {                    // inclusive Hillis-Steele scan
    unsigned long long v = threadIdx.x >= (unsigned)st ? part[threadIdx.x - st] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned long long a = part[threadIdx.x] - s;
  for (uint64_t i = lo; i < hi; i++) { uint64_t t = sums[i]; sums[i] = a; a += t; }
  if (threadIdx.x == 1023) *total = part[1023];
}
__global__ __launch_bounds__(BLK) void scan_apply(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ sums,
                                                  uint64_t* __restrict__ out, const uint64_t* __restrict__ total) {
  __shared__ unsigned long long part[BLK];
  uint64_t base = (uint64_t)blockIdx.x * 1024;
  uint32_t v[4];
  unsigned long long s = 0;
  for (int j = 0; j < 4; j++) { uint64_t i = base + threadIdx.x * 4 + j; v[j] = i < n ? in[i] : 0; s += v[j]; }
  // exclusive scan of the 256 thread sums: shuffles inside a wavefront, the four wavefront totals through LDS (one thread walking
  // all 256 partial sums was most of this kernel)
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += t; }
  if (lane == 63) part[wv] = inc;
  __syncthreads();
  unsigned long long wbase = 0;
  for (uint32_t w = 0; w < wv; w++) wbase += part[w];
  unsigned long long a = sums[blockIdx.x] + wbase + (inc - s);
  for (int j = 0; j < 4; j++) { uint64_t i = base + threadIdx.x * 4 + j; if (i < n) out[i] = a; a += v[j]; }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

__global__ __launch_bounds__(BLK) void compact_kernel(const uint64_t* __restrict__ run_keys, const uint32_t* __restrict__ run_counts,
                                                      const uint64_t* __restrict__ off1, const uint32_t* __restrict__ off2,
                                                      const uint32_t* __restrict__ ndist, const uint64_t* __restrict__ boff, int b2,
                                                      uint64_t n_buckets, uint64_t* __restrict__ keys, uint32_t* __restrict__ counts) {
  // one wave per bucket
  uint64_t b = ((uint64_t)blockIdx.x * BLK + threadIdx.x) / SHN_WAVE;
  if (b >= n_buckets) return;
  uint32_t lane = threadIdx.x & (SHN_WAVE - 1);
  uint32_t nd = ndist[b];
  if (!nd) return;
  uint64_t s0 = off1[b >> b2] + off2[b];
  uint64_t d0 = boff[b];
  for (uint32_t i = lane; i < nd; i += SHN_WAVE) { keys[d0 + i] = run_keys[s0 + i]; counts[d0 + i] = run_counts[s0 + i]; }
}

// ---------------------------------------------------------------- lookup
__global__ void lookup_kernel(TabIdx T, const uint32_t* __restrict__ tcounts, const uint64_t* __restrict__ q, uint64_t n,
                              uint32_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t j = shn_tab_find(T, q[i]);
  out[i] = j >= 0 ? tcounts[j] : 0u;
}

// ---------------------------------------------------------------- one-pass counting into a global hash table
// For inputs whose distinct k-mers are few next to the windows (deep coverage: 4.9 M of 750 M on BASELINE configs[1]) the
// partition pipeline above moves every window through HBM three times.  Here a block aggregates a tile of windows in an LDS
// hash table (heavy k-mers collapse there) and adds what is left -- (key, count) pairs -- to ONE open-addressing table in
// HBM with device-scope atomics (CAS on the key word, add on the count); the table's pairs then go through the pairs path
// (build_from_keys with counts) to become the same bucketed, sorted table.  A probe of the idea measured 14 ms against
// 25 ms (tools/probes/atomic_table_probe.hip).  The table is sized from the window count and grown on overflow; inputs
// with many distinct keys fall back to the partition pipeline.
#define DSLOTS 12288         // LDS slots per block (key 8 B + count 4 B = 144 KB of the 160 KB; 8192: +9 % time, 4096: +60 % in the probe)
__device__ __forceinline__ bool gtable_add(unsigned long long* __restrict__ gkeys, uint32_t* __restrict__ gcounts, uint64_t mask,
                                           uint64_t key, uint32_t c) {
  const unsigned long long kk = (unsigned long long)key + 1ULL;      // stored key + 1: 0 = empty slot (all-T at k = 32 would wrap: canonical counting never stores it, see host)
  uint64_t s = shn_mix64(key) & mask;
  for (int probe = 0; probe < 256; probe++) {
    unsigned long long cur = __hip_atomic_load(&gkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) { unsigned long long old = atomicCAS(&gkeys[s], 0ULL, kk); cur = old == 0 ? kk : old; }
    if (cur == kk) { atomicAdd(&gcounts[s], c); return true; }
    s = (s + 1) & mask;
  }
  return false;
}
template <bool CANON>
__global__ __launch_bounds__(1024) void count_direct_kernel(ReadsView v, int k, uint64_t n_tiles, unsigned long long* __restrict__ gkeys,
                                                            uint32_t* __restrict__ gcounts, uint64_t mask, uint32_t* __restrict__ overflow) {
  __shared__ unsigned long long lk[DSLOTS];
  __shared__ uint32_t lc[DSLOTS];
  __shared__ uint32_t stop;
  bool lost = false;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // (a table that turned out too small makes every insertion walk its whole probe limit: stop as soon as anybody lost a
    // window -- decided by one thread for the whole block, so that every wavefront takes the same way past the barriers)
    if (threadIdx.x == 0) stop = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j = threadIdx.x; j < DSLOTS; j += blockDim.x) { lk[j] = 0; lc[j] = 0; }
    __syncthreads();
    if (stop) break;
    const uint64_t r0 = tile * v.rt;
    const uint32_t nr = (uint32_t)min((uint64_t)v.rt, v.n_reads - r0);
    const uint32_t nid = nr * v.wmax;
    for (uint32_t i = threadIdx.x; i < nid; i += blockDim.x) {
      const uint32_t rl = i / v.wmax, pos = i - rl * v.wmax;
      uint64_t key;
      if (!gen_key<CANON>(v, r0 + rl, pos, k, key)) continue;
      const unsigned long long kk = (unsigned long long)key + 1ULL;
      uint32_t s = (uint32_t)(shn_mix64(key) >> 40) % DSLOTS;
      bool done = false;
#pragma unroll 1
      for (int probe = 0; probe < 8 && !done; probe++) {          // a few probes in LDS, then straight to the global table
        unsigned long long cur = lk[s];
        if (cur == 0) { unsigned long long old = atomicCAS(&lk[s], 0ULL, kk); cur = old == 0 ? kk : old; }
        if (cur == kk) { atomicAdd(&lc[s], 1u); done = true; }
        s = s + 1 == DSLOTS ? 0 : s + 1;
      }
      if (!done && !gtable_add(gkeys, gcounts, mask, key, 1)) { lost = true; atomicOr(overflow, 1u); }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < DSLOTS; j += blockDim.x)
      if (lk[j] && !gtable_add(gkeys, gcounts, mask, (uint64_t)(lk[j] - 1ULL), lc[j])) lost = true;
    __syncthreads();
  }
  if (lost) atomicOr(overflow, 1u);
}
__global__ void gtable_flag_kernel(const unsigned long long* __restrict__ gkeys, uint64_t slots, uint32_t* __restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots) flag[i] = gkeys[i] != 0 ? 1u : 0u;
}
__global__ void gtable_compact_kernel(const unsigned long long* __restrict__ gkeys, const uint32_t* __restrict__ gcounts, uint64_t slots,
                                      const uint64_t* __restrict__ pos, uint64_t* __restrict__ okeys, uint32_t* __restrict__ ocounts) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots && gkeys[i]) { okeys[pos[i]] = (uint64_t)(gkeys[i] - 1ULL); ocounts[pos[i]] = gcounts[i]; }
}

// ================================================================ host side
// Stages (the top-level entry points: count, extend, route, contig stage, unitigs ...) run one after the other on a GPU; a
// slot remembers the stage that used it last.  When an allocation fails, what the slots of EARLIER stages hold is given back
// and the allocation is tried again -- hipMalloc of tens of GB costs seconds here (tools/probes/malloc_probe.hip), so nothing
// is freed as long as everything fits.
// Who may use what, now that two batches can be in flight (the deferred back half of a step -- graph, sparse flow, merge -- on a
// second context and host thread beside the next step's front half) and the graph threads run on forked contexts: the
// stage slots (shn_ws(ctx): the default set or a context's own) belong to the top-level stages of ONE pipeline's front half only (count, extension, contig stage, probe
// table, routing, unitigs: one after the other on one host thread); everything the back half and the graph threads call (seed
// scans, the device scan, the LP batches) keeps its workspaces in its own context (shn_ctx::cws), which shn_ws_release_idle
// never touches.  get / release are serialised by g_ws_mu, so a slot's pointer and capacity are never written by two threads.
static std::mutex g_ws_mu;
// (single owner: the host thread that began the current stage of a set is the one that may ask for its slots until the set's next
// stage begins; anybody else asking is counted -- shn_debug_counter(2) -- and reported once: two threads on one slot would overwrite
// each other's buffers)
static uint64_t this_thread_tag() { return (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1ULL; }
static ShnWsSet g_default_wsset;               // per-process (one process per GPU)
static std::vector<ShnWsSet*> g_wssets;        // (g_ws_mu) the sets contexts own
ShnWsSet* shn_default_wsset() { return &g_default_wsset; }
ShnWsSet* shn_wsset_create() {
  ShnWsSet* s = new ShnWsSet();
  std::lock_guard<std::mutex> lk(g_ws_mu);
  g_wssets.push_back(s);
  return s;
}
void shn_wsset_destroy(ShnWsSet* s) {
  if (!s || s == &g_default_wsset) return;
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (size_t i = 0; i < g_wssets.size(); i++) if (g_wssets[i] == s) { g_wssets.erase(g_wssets.begin() + (ptrdiff_t)i); break; }
    for (auto& w : s->ws) if (w.p) { hipFree(w.p); w.p = nullptr; w.cap = 0; }
  }
  delete s;
}
void shn_stage_begin(shn_ctx* ctx) {
  ShnWsSet* set = ctx && ctx->wsset ? ctx->wsset : &g_default_wsset;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  set->stage++;
  set->stage_thread = this_thread_tag();
}
static size_t ws_release_idle_locked() {
  size_t freed = 0;
  auto sweep = [&](ShnWsSet* set) { for (auto& w : set->ws) if (w.p && w.stage < set->stage) { hipFree(w.p); freed += w.cap; w.p = nullptr; w.cap = 0; } };
  sweep(&g_default_wsset);
  for (ShnWsSet* set : g_wssets) sweep(set);
  return freed;
}
size_t shn_ws_release_idle() {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  return ws_release_idle_locked();
}
int ShnWs::get(size_t bytes, void** out) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  if (set) {
    const uint64_t owner = set->stage_thread;
    if (owner && owner != this_thread_tag()) {
      static bool told = false;
      shn_debug_count(2);
      if (!told) { told = true; fprintf(stderr, "[shannon_hip] stage workspace slot %d asked for by a thread that did not begin the current stage of its set\n", (int)(this - set->ws)); }
    }
    stage = set->stage;
  }
  bool grew = false;
  if (bytes > cap) {
    grew = true;
    if (p) hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                       // make room: cached blocks, then the slots of earlier stages
      (void)hipGetLastError();
      shn_dev_trim();
      ws_release_idle_locked();
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { p = nullptr; return shn_fail(SHN_ERR_NOMEM, std::string("hipMalloc workspace: ") + hipGetErrorString(e)); }
    cap = bytes;
  }
  *out = p;
  {
    // SHN_DEV_POISON: a slot that has just grown is filled with the poison byte; SHN_DEV_POISON_WS=1: on EVERY request (finds reads of
    // what an earlier call left behind: no call may rely on a slot's content across calls -- the seed scan's count / fetch pair,
    // which did until round 5, owns its offsets now, seeds.hip)
    static const bool every = getenv("SHN_DEV_POISON_WS") && getenv("SHN_DEV_POISON_WS")[0] == '1';
    if (every || grew) shn_poison(p, bytes, shn_current_stream());
  }
  return SHN_OK;
}
#define g_ws shn_ws(ctx)

int shn_device_scan_u32(shn_ctx* ctx, const uint32_t* d_in, uint64_t n, uint64_t* d_out /* n+1 */, uint64_t* total_host) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint64_t nblocks = cdiv(n, 1024);
  if (nblocks == 0) nblocks = 1;
  void* sums;
  int rc = ctx->cws[0].get((nblocks + 1) * 8, &sums);         // (per context: scans of different host threads / streams do not share it)
  if (rc) return rc;
  uint64_t* d_sums = (uint64_t*)sums;
  uint64_t* d_total = d_sums + nblocks;
  hipLaunchKernelGGL(scan_block_sums, dim3((uint32_t)nblocks), dim3(BLK), 0, s, d_in, n, d_sums);
  hipLaunchKernelGGL(scan_sums_serial, dim3(1), dim3(1024), 0, s, d_sums, nblocks, d_total);
  hipLaunchKernelGGL(scan_apply, dim3((uint32_t)nblocks), dim3(BLK), 0, s, d_in, n, d_sums, d_out, d_total);
  if (total_host) {                                          // (nullptr: the caller only needs out[] on the device -- no host sync)
    HIP_TRY(hipMemcpyAsync(total_host, d_total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return SHN_OK;
}

static ReadsView make_view(const shn_reads* r, int k) {
  ReadsView v;
  v.words = r->d_words; v.mask = r->d_mask; v.woff = r->d_woff; v.len = r->d_len;
  v.n_reads = r->n_reads; v.fixed_len = r->fixed_len; v.wpr = r->wpr;
  v.wmax = r->max_len >= (uint32_t)k ? r->max_len - k + 1 : 0;
  v.rt = v.wmax ? std::max<uint32_t>(1, TILE_IDS / v.wmax) : 1;
  v.has_n = r->n_invalid > 0;
  return v;
}

static int build_from_keys(shn_ctx* ctx, uint64_t* keysA, uint64_t* keysB, uint32_t* tmpc, uint32_t* cntA, uint32_t* cntB,
                           const std::vector<uint64_t>& off1h, int bits, int b2, int k, int canonical, uint64_t N,
                           uint64_t total, shn_table** out, bool* overflowed);

static int count_views(shn_ctx* ctx, const std::vector<ReadsView>& views, uint64_t upper, int k1, int both_strands, shn_table** out);

// reads [r0, r0 + n) of a view
static ReadsView sub_view(const ReadsView& v, uint64_t r0, uint64_t n) {
  ReadsView w = v;
  w.n_reads = n;
  if (v.woff) { w.woff = v.woff + r0; w.len = v.len + r0; }
  else { w.words = v.words + r0 * v.wpr; if (v.mask) w.mask = v.mask + r0 * v.wpr / 2; }     // (wpr is even: reads start on even word boundaries)
  return w;
}

extern "C" int shn_count_k1mers(shn_ctx* ctx, shn_reads* const* sets, int n_sets, int k1, int both_strands, shn_table** out) {
  if (!ctx || !sets || !out || n_sets <= 0) return shn_fail(SHN_ERR_ARG, "shn_count_k1mers: bad argument");
  if (k1 < 2 || k1 > 32) return shn_fail(SHN_ERR_ARG, "shn_count_k1mers: k1 must be in [2,32]");
  SHN_ENTER(ctx);
  shn_stage_begin(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion ttot(ctx, T_COUNT_TOTAL);
  uint64_t upper = 0;
  std::vector<ReadsView> views;
  for (int i = 0; i < n_sets; i++) {
    if (!sets[i]) return shn_fail(SHN_ERR_ARG, "shn_count_k1mers: NULL read set");
    ReadsView v = make_view(sets[i], k1);
    views.push_back(v);
    upper += v.n_reads * (uint64_t)v.wmax;
  }
  // The partition pipeline holds two 8-byte key buffers + one 4-byte count buffer for all windows at once: 20 B x 7.5 G windows
  // = 150 GB at 100 M reads, half the device for the whole run (hipMalloc of such blocks costs seconds, so they are never given
  // back).  Beyond 2^31 windows the reads are counted in chunks of at most that many windows, each chunk to its own table, and
  // the tables' (key, count) pairs are reduced by key (the pairs path of the exchange step) -- a few GB of extra traffic.
  uint64_t chunk_windows = 1ULL << 31;
  if (getenv("SHN_COUNT_CHUNK")) chunk_windows = std::max<uint64_t>(1, strtoull(getenv("SHN_COUNT_CHUNK"), nullptr, 10));   // (tests)
  if (upper <= chunk_windows) return count_views(ctx, views, upper, k1, both_strands, out);
  {
    // inputs of this size go through the super-k-mer path in one piece (its records are a quarter of the windows' keys); what it
    // does not take (k1 < 20, forward counting at k1 = 32) is counted in chunks below
    int handled = 0;
    int rc = shn_count_superkmers(ctx, views, upper, k1, both_strands, out, &handled);
    if (rc || handled) return rc;
  }
  std::vector<shn_table*> parts;
  auto drop = [&]() { for (shn_table* t : parts) shn_table_destroy(t); parts.clear(); };
  {
    std::vector<ReadsView> cur;
    uint64_t cur_w = 0;
    auto flush = [&]() -> int {
      if (cur.empty()) return SHN_OK;
      shn_table* t = nullptr;
      int rc = count_views(ctx, cur, cur_w, k1, both_strands, &t);
      if (rc) return rc;
      parts.push_back(t);
      cur.clear(); cur_w = 0;
      return SHN_OK;
    };
    for (const ReadsView& v : views) {
      if (!v.wmax || !v.n_reads) continue;
      uint64_t r0 = 0;
      while (r0 < v.n_reads) {
        const uint64_t room = (chunk_windows - cur_w) / v.wmax;
        if (room == 0) { int rc = flush(); if (rc) { drop(); return rc; } continue; }
        const uint64_t n = std::min<uint64_t>(v.n_reads - r0, std::max<uint64_t>(room, 1));
        cur.push_back(sub_view(v, r0, n));
        cur_w += n * v.wmax;
        r0 += n;
        if (cur_w >= chunk_windows) { int rc = flush(); if (rc) { drop(); return rc; } }
      }
    }
    int rc = flush();
    if (rc) { drop(); return rc; }
  }
  uint64_t np = 0;
  for (shn_table* t : parts) np += t->n;
  uint64_t* pk = nullptr; uint32_t* pc = nullptr;
  if (shn_dev_malloc(&pk, (np + 1) * 8) != hipSuccess || shn_dev_malloc(&pc, (np + 1) * 4) != hipSuccess) {
    shn_dev_free(pk); shn_dev_free(pc); drop();
    return shn_fail(SHN_ERR_NOMEM, "shn_count_k1mers: out of device memory for the chunk tables");
  }
  uint64_t at = 0;
  for (shn_table* t : parts) {
    if (t->n) {
      hipMemcpyAsync(pk + at, t->d_keys, t->n * 8, hipMemcpyDeviceToDevice, s);
      hipMemcpyAsync(pc + at, t->d_counts, t->n * 4, hipMemcpyDeviceToDevice, s);
    }
    at += t->n;
  }
  hipStreamSynchronize(s);
  drop();
  int rc = shn_table_from_pairs(ctx, pk, pc, np, k1, both_strands ? 1 : 0, out);
  hipStreamSynchronize(s);
  shn_dev_free(pk); shn_dev_free(pc);
  return rc;
}

static int count_views(shn_ctx* ctx, const std::vector<ReadsView>& views, uint64_t upper, int k1, int both_strands, shn_table** out) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  // One-pass path first (see count_direct_kernel): worth it when the table is small next to the windows; the size is a
  // guess (windows / 32, at most 2^24 slots to begin with) corrected by what earlier calls of this process needed.
  {
    int& learned_log2 = ctx->count_direct_log2;            // slots that sufficed last time on this context (0: none yet; -1: the
                                                           // one-pass path gave up on this context's input -- do not try again)
    const int mode = getenv("SHN_COUNT_DIRECT") ? atoi(getenv("SHN_COUNT_DIRECT")) : 1;      // 0 off, 1 large inputs, 2 always (tests)
    // (stored key + 1: k1 = 32 is fine for canonical counting -- the one key that would wrap, all-T, is never the canonical form of
    // a pair, its reverse complement all-A is)
    const bool want = mode != 0 && (upper >= (1ULL << 22) || mode == 2) && (k1 < 32 || both_strands) && upper > 0 && (learned_log2 >= 0 || mode == 2);
    int lg = learned_log2 > 0 ? learned_log2 : 0;
    if (!lg) { lg = 20; while (lg < 24 && (1ULL << lg) < upper / 32) lg++; }
    if (getenv("SHN_COUNT_DIRECT_LOG2")) lg = atoi(getenv("SHN_COUNT_DIRECT_LOG2"));          // (tests: start too small, grow)
    for (int attempt = 0; want && attempt < 8 && lg <= 27; attempt++) {
      const uint64_t slots = 1ULL << lg;
      unsigned long long* gk = nullptr; uint32_t* gc = nullptr; uint32_t* flag = nullptr; uint64_t* pos = nullptr; uint32_t* d_ov = nullptr;
      uint64_t* pk = nullptr; uint32_t* pcn = nullptr;
      auto freeall = [&]() { if (gk) shn_dev_free(gk); if (gc) shn_dev_free(gc); if (flag) shn_dev_free(flag); if (pos) shn_dev_free(pos);
                             if (d_ov) shn_dev_free(d_ov); if (pk) shn_dev_free(pk); if (pcn) shn_dev_free(pcn); };
#define TRYD(x) do { hipError_t _e = (x); if (_e != hipSuccess) { freeall(); return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
      TRYD(shn_dev_malloc(&gk, slots * 8)); TRYD(shn_dev_malloc(&gc, slots * 4)); TRYD(shn_dev_malloc(&flag, (slots + 1) * 4));
      TRYD(shn_dev_malloc(&pos, (slots + 2) * 8)); TRYD(shn_dev_malloc(&d_ov, 64));
      TRYD(hipMemsetAsync(gk, 0, slots * 8, s)); TRYD(hipMemsetAsync(gc, 0, slots * 4, s)); TRYD(hipMemsetAsync(d_ov, 0, 64, s));
      for (auto& v : views) {
        if (!v.wmax || !v.n_reads) continue;
        TimerRegion t(ctx, T_COUNT_DIRECT);
        uint64_t n_tiles = cdiv(v.n_reads, v.rt);
        uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, 2048);
        if (both_strands) hipLaunchKernelGGL(count_direct_kernel<true>, dim3(grid), dim3(1024), 0, s, v, k1, n_tiles, gk, gc, slots - 1, d_ov);
        else hipLaunchKernelGGL(count_direct_kernel<false>, dim3(grid), dim3(1024), 0, s, v, k1, n_tiles, gk, gc, slots - 1, d_ov);
      }
      hipLaunchKernelGGL(gtable_flag_kernel, dim3((uint32_t)cdiv(slots, 256)), dim3(256), 0, s, gk, slots, flag);
      uint64_t nd = 0;
      int rcd = shn_device_scan_u32(ctx, flag, slots, pos, &nd);
      if (rcd) { freeall(); return rcd; }
      uint32_t ov = 0;
      TRYD(hipMemcpy(&ov, d_ov, 4, hipMemcpyDeviceToHost));
      if (ov || nd * 10 > slots * 6) {                      // too full (long probe chains) or lost windows: a bigger table, or give up
        freeall();
        if (mode != 2 && nd * 10 > slots * 6 && !ov && (nd << 5) > upper) { learned_log2 = -1; break; }      // many distinct keys for the windows: the partition pipeline is the better tool
        if (mode != 2 && lg + (ov ? 3 : 2) > 27) learned_log2 = -1;
        lg += ov ? 3 : 2;
        continue;
      }
      TRYD(shn_dev_malloc(&pk, (nd + 1) * 8)); TRYD(shn_dev_malloc(&pcn, (nd + 1) * 4));
      hipLaunchKernelGGL(gtable_compact_kernel, dim3((uint32_t)cdiv(slots, 256)), dim3(256), 0, s, gk, gc, slots, pos, pk, pcn);
      shn_table* tb = nullptr;
      int rcp = shn_table_from_pairs(ctx, pk, pcn, nd, k1, both_strands ? 1 : 0, &tb);
      { hipError_t es = hipStreamSynchronize(s);
        if (es != hipSuccess) { freeall(); if (!rcp) shn_table_destroy(tb); return shn_fail(SHN_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(es)); } }
      freeall();
#undef TRYD
      if (rcp) return rcp;
      if (!getenv("SHN_COUNT_DIRECT_LOG2")) learned_log2 = lg;
      *out = tb;
      return SHN_OK;
    }
  }
  {
    int handled = 0;
    int rc = shn_count_superkmers(ctx, views, upper, k1, both_strands, out, &handled);
    if (rc || handled) return rc;
  }
  // bits_n: enough buckets if every window were a distinct key (the histogram pass runs at this resolution);
  // the final number of buckets follows the HyperLogLog estimate of the distinct keys and only grows on overflow
  int bits_n = 0;
  while (bits_n < 23 && (upper >> bits_n) > TARGET_BUCKET) bits_n++;
  int bits = -1;                                  // chosen after the first histogram pass
  int bits_hist = -1;                             // resolution of the histogram held in h1_fine
  std::vector<uint64_t> h1_fine;
  for (int attempt = 0; attempt < 4; attempt++) {
    int rc;
    void* p;
    if (bits < 0 || bits > bits_hist) {
      // (re)compute the level-1 histogram (+ sketch) at resolution max(bits, bits_n)
      bits_hist = std::max(bits, bits_n);
      const int hb1 = (bits_hist + 1) / 2, hb2 = bits_hist - hb1;
      const int hnb1 = 1 << hb1;
      if ((rc = g_ws[0].get((size_t)hnb1 * 16 + 64 + (4u << HLL_BITS), &p))) return rc;
      unsigned long long* d_h = (unsigned long long*)p;
      uint32_t* d_hll = (uint32_t*)(d_h + 2 * (size_t)hnb1 + 8);
      HIP_TRY(hipMemsetAsync(d_h, 0, (size_t)hnb1 * 8, s));
      HIP_TRY(hipMemsetAsync(d_hll, 0, 4u << HLL_BITS, s));
      for (auto& v : views) {
        if (!v.wmax || !v.n_reads) continue;
        TimerRegion t(ctx, T_HIST1);                 // one region per launch (bench.py compares with rocprofv3 per kernel)
        uint64_t n_tiles = cdiv(v.n_reads, v.rt);
        uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, 2048);
        if (both_strands) hipLaunchKernelGGL(hist1_kernel<true>, dim3(grid), dim3(BLK), hnb1 * 4, s, v, k1, bits_hist, hb2, n_tiles, d_h, d_hll);
        else hipLaunchKernelGGL(hist1_kernel<false>, dim3(grid), dim3(BLK), hnb1 * 4, s, v, k1, bits_hist, hb2, n_tiles, d_h, d_hll);
      }
      h1_fine.resize(hnb1);
      std::vector<uint32_t> regs(1u << HLL_BITS);
      HIP_TRY(hipMemcpyAsync(h1_fine.data(), d_h, (size_t)hnb1 * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(regs.data(), d_hll, 4u << HLL_BITS, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (bits < 0) {
        const double m = (double)(1u << HLL_BITS);
        double z = 0;
        uint32_t zeros = 0;
        for (uint32_t rg : regs) { z += std::ldexp(1.0, -(int)rg); zeros += rg == 0; }
        double est = (0.7213 / (1.0 + 1.079 / m)) * m * m / z;
        if (est < 2.5 * m && zeros) est = m * std::log(m / (double)zeros);       // small-range correction
        // ~300 distinct keys per bucket (a bucket's LDS table takes CAP_LIMIT = 950), 25 % head room on the
        // estimate; never more buckets than the all-distinct rule gives, never fewer than 2^10 when there is work
        const double want = est * 1.25 / 300.0;
        bits = 0;
        while (bits < bits_n && (double)(1ULL << bits) < want) bits++;
        if (upper > (1ULL << 20)) bits = std::max(bits, std::min(bits_n, 10));
      }
    }
    int b1 = (bits + 1) / 2;
    if (bits - b1 > 15) b1 = bits - 15;
    const int b2 = bits - b1;
    const int nb1 = 1 << b1;
    // level-1 digits are prefixes of the finer ones: fold the fine histogram
    const int hb1 = (bits_hist + 1) / 2;
    std::vector<uint64_t> h1(nb1, 0), off1(nb1 + 1);
    for (size_t i = 0; i < h1_fine.size(); i++) h1[i >> (hb1 - b1)] += h1_fine[i];
    if ((rc = g_ws[0].get((size_t)(1 << hb1) * 16 + 64 + (4u << HLL_BITS), &p))) return rc;
    unsigned long long* d_cursor1 = (unsigned long long*)p + (size_t)(1 << hb1);
    uint64_t N = 0;
    for (int i = 0; i < nb1; i++) { off1[i] = N; N += h1[i]; }
    off1[nb1] = N;
    HIP_TRY(hipMemcpyAsync(d_cursor1, off1.data(), (size_t)nb1 * 8, hipMemcpyHostToDevice, s));
    void *pa, *pb, *pc;
    if ((rc = g_ws[1].get((N + 2) * 8, &pa))) return rc;
    if ((rc = g_ws[2].get((N + 2) * 8, &pb))) return rc;
    if ((rc = g_ws[3].get((N + 2) * 4, &pc))) return rc;
    uint64_t* keysA = (uint64_t*)pa;
    uint64_t* keysB = (uint64_t*)pb;
    {
      for (auto& v : views) {
        if (!v.wmax || !v.n_reads) continue;
        TimerRegion t(ctx, T_SCATTER1);
        uint64_t n_tiles = cdiv(v.n_reads, v.rt);
        uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, 4096);   // (2048: +0.7 ms; 512: +3 ms -- occupancy, not L2 write combining, is what matters)
        size_t sh = (size_t)nb1 * 4 + (size_t)nb1 * 8;
        if (both_strands) hipLaunchKernelGGL(scatter1_kernel<true>, dim3(grid), dim3(BLK), sh, s, v, k1, bits, b2, n_tiles, d_cursor1, keysA);
        else hipLaunchKernelGGL(scatter1_kernel<false>, dim3(grid), dim3(BLK), sh, s, v, k1, bits, b2, n_tiles, d_cursor1, keysA);
      }
    }
    bool ov = false;
    rc = build_from_keys(ctx, keysA, keysB, (uint32_t*)pc, nullptr, nullptr, off1, bits, b2, k1, both_strands ? 1 : 0, N, N, out, &ov);
    if (rc) return rc;
    if (!ov) return SHN_OK;
    shn_table_destroy(*out);
    *out = nullptr;
    bits = std::min(23, bits + 2);
  }
  return shn_fail(SHN_ERR_OVERFLOW, "shn_count_k1mers: a final bucket exceeded the LDS hash capacity after 4 attempts");
}

// keysA holds N keys partitioned at level 1 (offsets off1h); produces the dense table.
// cntA/cntB non-NULL: keys carry counts (pairs path); `total` = number of windows represented.
static int build_from_keys(shn_ctx* ctx, uint64_t* keysA, uint64_t* keysB, uint32_t* tmpc, uint32_t* cntA, uint32_t* cntB,
                           const std::vector<uint64_t>& off1h, int bits, int b2, int k, int canonical, uint64_t N,
                           uint64_t total, shn_table** out, bool* overflowed) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  int b1 = bits - b2;
  int nb1 = 1 << b1, nb2 = 1 << b2;
  uint64_t nbk = (uint64_t)1 << bits;
  uint64_t max1 = 0;
  for (int i = 0; i < nb1; i++) max1 = std::max(max1, off1h[i + 1] - off1h[i]);
  if (max1 >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_OVERFLOW, "level-1 partition larger than 2^32 keys");
  void* p;
  int rc;
  // below level 1: one level of b2 bits while that is at most 256 streams... up to 2^11 for small inputs (their tiles are short
  // either way); else a middle level and a last level of at most 8 bits each (bits <= 23, b1 >= 8 there: b2 <= 15)
  const bool three = b2 > 11;
  const int b3 = three ? b2 / 2 : b2, bm = b2 - b3;              // last level, middle level (bm = 0: none)
  const uint32_t nbm = 1u << bm, nb3 = 1u << b3;
  const uint32_t n_seg3 = (uint32_t)nb1 * nbm;                    // segments of the last level
  size_t need = (size_t)(nb1 + 1) * 8 + nbk * 4 * 4 + 64 + (nbk + 2) * 8 + (three ? ((size_t)n_seg3 + 2) * 8 + (size_t)n_seg3 * 12 : 0);
  if ((rc = g_ws[4].get(need, &p))) return rc;
  uint64_t* d_off1 = (uint64_t*)p;
  uint64_t* d_boff = d_off1 + (nb1 + 1);
  uint32_t* d_hist2 = (uint32_t*)(d_boff + nbk + 2);
  uint32_t* d_off2 = d_hist2 + nbk;
  uint32_t* d_cursor2 = d_off2 + nbk;
  uint32_t* d_ndist = d_cursor2 + nbk;
  uint32_t* d_ovf = d_ndist + nbk;
  uint64_t* d_seg3 = (uint64_t*)(d_ovf + 16);                     // (three levels) absolute starts of the last level's segments
  uint32_t* d_histm = (uint32_t*)(d_seg3 + n_seg3 + 2);
  uint32_t* d_offm = d_histm + n_seg3;
  uint32_t* d_cursorm = d_offm + n_seg3;
  HIP_TRY(hipMemcpyAsync(d_off1, off1h.data(), (size_t)(nb1 + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_hist2, 0, nbk * 4, s));
  HIP_TRY(hipMemsetAsync(d_ovf, 0, 4, s));
  const uint32_t tile2 = TILE_KEYS;
  const size_t pad2 = 0;
  const uint64_t *segs = d_off1;                                  // segments of the last level and where their keys are
  uint32_t n_segs = (uint32_t)nb1;
  uint64_t max_seg = max1;
  uint64_t *kin = keysA, *kout = keysB;
  uint32_t *cin_ = cntA, *cout_ = cntB;
  if (three) {
    HIP_TRY(hipMemsetAsync(d_histm, 0, (size_t)n_seg3 * 4, s));
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, cdiv(max1, tile2));
    {
      TimerRegion t(ctx, T_HIST2);
      hipLaunchKernelGGL(hist_keys_kernel, dim3(tiles, (uint32_t)nb1), dim3(BLK), nbm * 4, s, kin, d_off1, (uint32_t)nb1, bits, b3, nbm, d_histm, tile2);
    }
    hipLaunchKernelGGL(scan_rows_kernel, dim3((uint32_t)nb1), dim3(BLK), 0, s, d_histm, (int)nbm, d_offm, d_cursorm);
    {
      TimerRegion t(ctx, T_SCATTER2);
      if (cntA) hipLaunchKernelGGL((scatter_keys_kernel<true>), dim3(tiles, (uint32_t)nb1), dim3(BLK), nbm * 8 + pad2, s, kin, cin_, d_off1, (uint32_t)nb1, bits, b3, nbm, d_cursorm, kout, cout_, tile2);
      else hipLaunchKernelGGL((scatter_keys_kernel<false>), dim3(tiles, (uint32_t)nb1), dim3(BLK), nbm * 8 + pad2, s, kin, nullptr, d_off1, (uint32_t)nb1, bits, b3, nbm, d_cursorm, kout, nullptr, tile2);
    }
    hipLaunchKernelGGL(next_segments_kernel, dim3((uint32_t)cdiv((uint64_t)n_seg3 + 1, 256)), dim3(256), 0, s, d_off1, (uint32_t)nb1, d_offm, nbm, d_seg3);
    // the largest segment of the last level sizes its grid
    std::vector<uint32_t> hm(n_seg3);
    HIP_TRY(hipMemcpyAsync(hm.data(), d_histm, (size_t)n_seg3 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    max_seg = 0;
    for (uint32_t v : hm) max_seg = std::max<uint64_t>(max_seg, v);
    segs = d_seg3; n_segs = n_seg3;
    std::swap(kin, kout); std::swap(cin_, cout_);
  }
  {
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, cdiv(max_seg, tile2));
    const uint32_t gy = std::min<uint32_t>(n_segs, 32768u);
    {
      TimerRegion t(ctx, T_HIST2);
      hipLaunchKernelGGL(hist_keys_kernel, dim3(tiles, gy), dim3(BLK), nb3 * 4, s, kin, segs, n_segs, bits, 0, nb3, d_hist2, tile2);
    }
    hipLaunchKernelGGL(scan_rows_kernel, dim3(n_segs), dim3(BLK), 0, s, d_hist2, (int)nb3, d_off2, d_cursor2);
    {
      TimerRegion t(ctx, T_SCATTER2);
      if (cntA) hipLaunchKernelGGL((scatter_keys_kernel<true>), dim3(tiles, gy), dim3(BLK), nb3 * 8 + pad2, s, kin, cin_, segs, n_segs, bits, 0, nb3, d_cursor2, kout, cout_, tile2);
      else hipLaunchKernelGGL((scatter_keys_kernel<false>), dim3(tiles, gy), dim3(BLK), nb3 * 8 + pad2, s, kin, nullptr, segs, n_segs, bits, 0, nb3, d_cursor2, kout, nullptr, tile2);
    }
    std::swap(kin, kout); std::swap(cin_, cout_);
  }
  // kin: the keys grouped by final bucket; kout / tmpc take the (key, count) runs of the buckets
  {
    TimerRegion t(ctx, T_COUNT);
    if (cntA) hipLaunchKernelGGL(buckets_kernel<true>, dim3((uint32_t)nbk), dim3(BLK), 0, s, kin, cin_, segs, d_off2, d_hist2, b3, kout, tmpc, d_ndist, d_ovf);
    else hipLaunchKernelGGL(buckets_kernel<false>, dim3((uint32_t)nbk), dim3(BLK), 0, s, kin, nullptr, segs, d_off2, d_hist2, b3, kout, tmpc, d_ndist, d_ovf);
  }
  uint64_t D = 0;
  shn_table* t = new shn_table();
  memset(t, 0, sizeof(*t));
  t->ctx = ctx; t->device = ctx->device; t->k = k; t->canonical = canonical; t->bits = bits; t->n_buckets = nbk; t->total = total;
  {
    TimerRegion tr(ctx, T_COMPACT);
    if ((rc = shn_device_scan_u32(ctx, d_ndist, nbk, d_boff, &D))) { delete t; return rc; }
    uint32_t ovf = 0;
    { hipError_t e1 = hipMemcpyAsync(&ovf, d_ovf, 4, hipMemcpyDeviceToHost, s);
      if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
      if (e1 != hipSuccess) { delete t; return shn_fail(SHN_ERR_HIP, std::string("build_from_keys: ") + hipGetErrorString(e1)); } }
    *overflowed = ovf != 0;
    t->n = D;
#define TRYT(x) do { hipError_t _e = (x); if (_e != hipSuccess) { shn_table_destroy(t); return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
    TRYT(shn_dev_malloc(&t->d_keys, (D + 1) * 8));
    TRYT(shn_dev_malloc(&t->d_counts, (D + 1) * 4));
    TRYT(shn_dev_malloc(&t->d_bucket_off, (nbk + 1) * 8));
    TRYT(hipMemcpyAsync(t->d_bucket_off, d_boff, (nbk + 1) * 8, hipMemcpyDeviceToDevice, s));
#undef TRYT
    if (!ovf && D) {
      uint32_t blocks = (uint32_t)cdiv(nbk * SHN_WAVE, BLK);
      hipLaunchKernelGGL(compact_kernel, dim3(blocks), dim3(BLK), 0, s, kout, tmpc, segs, d_off2, d_ndist, d_boff, b3, nbk, t->d_keys, t->d_counts);
    }
  }
  HIP_TRY(hipGetLastError());
  *out = t;
  return SHN_OK;
}

extern "C" void shn_table_destroy(shn_table* t) {
  if (!t) return;
  hipSetDevice(t->device);
  shn_dev_free(t->d_keys);
  shn_dev_free(t->d_counts);
  shn_dev_free(t->d_bucket_off);
  delete t;
}
extern "C" uint64_t shn_table_size(const shn_table* t) { return t ? t->n : 0; }
extern "C" uint64_t shn_table_total(const shn_table* t) { return t ? t->total : 0; }
extern "C" int shn_table_k(const shn_table* t) { return t ? t->k : 0; }
extern "C" int shn_table_canonical(const shn_table* t) { return t ? t->canonical : 0; }

extern "C" int shn_table_download(shn_ctx* ctx, const shn_table* t, uint64_t* keys, uint32_t* counts) {
  if (!ctx || !t) return shn_fail(SHN_ERR_ARG, "shn_table_download: NULL argument");
  SHN_ENTER(ctx);
  if (keys) HIP_TRY(hipMemcpyAsync(keys, t->d_keys, t->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (counts) HIP_TRY(hipMemcpyAsync(counts, t->d_counts, t->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SHN_OK;
}

extern "C" int shn_table_dump(shn_ctx* ctx, const shn_table* t, uint32_t lower, uint64_t* keys, uint32_t* counts, uint64_t* n) {
  if (!ctx || !t || !n) return shn_fail(SHN_ERR_ARG, "shn_table_dump: NULL argument");
  std::vector<uint64_t> hk(t->n);
  std::vector<uint32_t> hc(t->n);
  int rc = shn_table_download(ctx, t, hk.data(), hc.data());
  if (rc) return rc;
  std::vector<std::pair<uint64_t, uint32_t>> v;
  v.reserve(t->canonical ? 2 * t->n : t->n);
  for (uint64_t i = 0; i < t->n; i++) {
    uint64_t key = hk[i];
    uint64_t c = hc[i];
    if (t->canonical) {
      uint64_t rc2 = shn_revcomp_host(key, t->k);
      if (rc2 == key) { c *= 2; if (c >= lower) v.push_back({key, (uint32_t)std::min<uint64_t>(c, 0xFFFFFFFFULL)}); }
      else if (c >= lower) { v.push_back({key, (uint32_t)c}); v.push_back({rc2, (uint32_t)c}); }
    } else if (c >= lower) v.push_back({key, (uint32_t)c});
  }
  if (!keys || !counts) { *n = v.size(); return SHN_OK; }
  if (*n < v.size()) return shn_fail(SHN_ERR_ARG, "shn_table_dump: output arrays too small");
  std::sort(v.begin(), v.end());
  for (size_t i = 0; i < v.size(); i++) { keys[i] = v[i].first; counts[i] = v[i].second; }
  *n = v.size();
  return SHN_OK;
}

// ---- `jellyfish dump -L lower` as a table (shannon.py:237-241, 441): the k1-mers whose count in the reference's (strand-doubled) input
// is below `lower` never enter k1mer.dict_org, so no later stage sees them.  A stable compaction on the device: the survivors keep
// their order (bucket, then key), bucket b of the new table starts where the scan of the keep flags stands at b's old start, layout,
// bits and minimizer length are the old table's.  A canonical table holds count(x) = count(rc x) of the doubled input once; a k1-mer
// that is its own reverse complement stands for twice its stored count there (the rule of shn_table_dump).
__global__ void table_keep_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, uint64_t n, int k, int canonical,
                                  uint32_t lower, uint32_t* __restrict__ flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t c = counts[i];
    if (canonical && shn_revcomp(keys[i], k) == keys[i]) c *= 2;
    flag[i] = c >= lower ? 1u : 0u;
  }
}
__global__ void table_keep_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ flag,
                                          const uint64_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ okeys, uint32_t* __restrict__ ocounts) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (flag[i]) { const uint64_t o = pos[i]; okeys[o] = keys[i]; ocounts[o] = counts[i]; }
}
__global__ void table_keep_boff_kernel(const uint64_t* __restrict__ old_off, uint64_t nb1, const uint64_t* __restrict__ pos, uint64_t* __restrict__ new_off) {
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb1; b += (uint64_t)gridDim.x * blockDim.x) new_off[b] = pos[old_off[b]];
}
extern "C" int shn_table_filter_lower(shn_ctx* ctx, const shn_table* t, uint32_t lower, shn_table** out) {
  if (!ctx || !t || !out) return shn_fail(SHN_ERR_ARG, "shn_table_filter_lower: NULL argument");
  if (t->n >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_table_filter_lower: more than 2^32 keys");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t n = t->n, nb1 = t->n_buckets + 1;
  ShnDevBufs bufs(s);
  uint32_t* d_flag = nullptr; uint64_t* d_pos = nullptr;
  HIP_TRY(bufs.get(&d_flag, (n + 1) * 4)); HIP_TRY(bufs.get(&d_pos, (n + 2) * 8));
  const dim3 grid((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cdiv(n, 256), 1u << 20))), blk(256);
  if (n) hipLaunchKernelGGL(table_keep_kernel, grid, blk, 0, s, (const uint64_t*)t->d_keys, (const uint32_t*)t->d_counts, n, t->k, t->canonical, lower, d_flag);
  uint64_t kept = 0;
  if (n) { int rc = shn_device_scan_u32(ctx, d_flag, n, d_pos, &kept); if (rc) return rc; }
  else HIP_TRY(hipMemsetAsync(d_pos, 0, 16, s));
  shn_table* o = new shn_table();
  *o = *t;
  o->ctx = ctx; o->n = kept; o->d_keys = nullptr; o->d_counts = nullptr; o->d_bucket_off = nullptr;
#define TRYT(x) do { hipError_t _e = (x); if (_e != hipSuccess) { shn_table_destroy(o); return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
  TRYT(shn_dev_malloc(&o->d_keys, (kept + 1) * 8));
  TRYT(shn_dev_malloc(&o->d_counts, (kept + 1) * 4));
  TRYT(shn_dev_malloc(&o->d_bucket_off, nb1 * 8));
  if (n) hipLaunchKernelGGL(table_keep_scatter_kernel, grid, blk, 0, s, (const uint64_t*)t->d_keys, (const uint32_t*)t->d_counts, (const uint32_t*)d_flag,
                            (const uint64_t*)d_pos, n, o->d_keys, o->d_counts);
  hipLaunchKernelGGL(table_keep_boff_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(nb1, 256), 1u << 20)), blk, 0, s, (const uint64_t*)t->d_bucket_off, nb1,
                     (const uint64_t*)d_pos, o->d_bucket_off);
  TRYT(hipGetLastError());
  TRYT(hipStreamSynchronize(s));
#undef TRYT
  *out = o;
  return SHN_OK;
}

extern "C" int shn_table_device_ptrs(const shn_table* t, void** keys, void** counts) {
  if (!t) return shn_fail(SHN_ERR_ARG, "shn_table_device_ptrs: NULL table");
  if (keys) *keys = t->d_keys;
  if (counts) *counts = t->d_counts;
  return SHN_OK;
}

extern "C" int shn_table_lookup(shn_ctx* ctx, const shn_table* t, const uint64_t* keys, uint64_t n, uint32_t* counts) {
  if (!ctx || !t || (!keys && n) || (!counts && n)) return shn_fail(SHN_ERR_ARG, "shn_table_lookup: NULL argument");
  if (!n) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint64_t* dq; uint32_t* dc;
  HIP_TRY(shn_hip_malloc(&dq, n * 8));
  HIP_TRY(shn_hip_malloc(&dc, n * 4));
  HIP_TRY(hipMemcpyAsync(dq, keys, n * 8, hipMemcpyHostToDevice, s));
  {
    TimerRegion tr(ctx, T_LOOKUP);
    hipLaunchKernelGGL(lookup_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, shn_tab_idx(t), t->d_counts, dq, n, dc);
  }
  HIP_TRY(hipMemcpyAsync(counts, dc, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  hipFree(dq); hipFree(dc);
  return SHN_OK;
}

// ---------------------------------------------------------------- (key,count) pairs path + sharding
__global__ void sum_counts_kernel(const uint32_t* __restrict__ c, uint64_t n, unsigned long long* __restrict__ out) {
  unsigned long long a = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a += c[i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((threadIdx.x & 63) == 0 && a) atomicAdd(out, a);
}

// ---- the table of a strand-specific paired run: the k1-mers of reads_1 as they are and those of RC(reads_2) = the reverse complements
// of the forward k1-mers of reads_2 (shannon.py:407-411, :436-439 without -C), equal keys summed -- from the two forward tables, on
// the device (the host form downloaded both tables, reverse-complemented one in numpy and uploaded the concatenation)
__global__ void table_rc_concat_kernel(const uint64_t* __restrict__ ka, const uint32_t* __restrict__ ca, uint64_t na, const uint64_t* __restrict__ kb,
                                       const uint32_t* __restrict__ cb, uint64_t nb, int k, uint64_t* __restrict__ keys, uint32_t* __restrict__ cnts) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i < na) { keys[i] = ka[i]; cnts[i] = ca[i]; }
    else { keys[i] = shn_revcomp(kb[i - na], k); cnts[i] = cb[i - na]; }
  }
}
extern "C" int shn_table_from_pairs(shn_ctx* ctx, const void* dev_keys, const void* dev_counts, uint64_t n, int k1, int canonical, shn_table** out);
extern "C" int shn_table_merge_rc(shn_ctx* ctx, const shn_table* fwd, const shn_table* other, shn_table** out) {
  if (!ctx || !fwd || !other || !out) return shn_fail(SHN_ERR_ARG, "shn_table_merge_rc: NULL argument");
  if (fwd->canonical || other->canonical || fwd->k != other->k) return shn_fail(SHN_ERR_ARG, "shn_table_merge_rc: two plain tables of one k");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t n = fwd->n + other->n;
  ShnDevBufs bufs(s);
  uint64_t* d_k = nullptr; uint32_t* d_c = nullptr;
  HIP_TRY(bufs.get(&d_k, (n + 1) * 8)); HIP_TRY(bufs.get(&d_c, (n + 1) * 4));
  if (n) hipLaunchKernelGGL(table_rc_concat_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n, 256), 1u << 20)), dim3(256), 0, s, (const uint64_t*)fwd->d_keys,
                            (const uint32_t*)fwd->d_counts, fwd->n, (const uint64_t*)other->d_keys, (const uint32_t*)other->d_counts, other->n, fwd->k, d_k, d_c);
  HIP_TRY(hipGetLastError());
  int rc = shn_table_from_pairs(ctx, d_k, d_c, n, fwd->k, 0, out);
  if (rc) return rc;
  (*out)->total = fwd->total + other->total;
  return SHN_OK;
}

extern "C" int shn_table_from_pairs(shn_ctx* ctx, const void* dev_keys, const void* dev_counts, uint64_t n, int k1,
                                    int canonical, shn_table** out) {
  if (!ctx || !out || (n && (!dev_keys || !dev_counts))) return shn_fail(SHN_ERR_ARG, "shn_table_from_pairs: NULL argument");
  if (n >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_table_from_pairs: more than 2^32 pairs");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion ttot(ctx, T_TABLE_BUILD);           // small tables (probe / seed / exchanged shards): one region, kept apart
  TimingOff toff(ctx);                            // from the per-kernel timers of the counting pass
  const uint64_t* keys = (const uint64_t*)dev_keys;
  const uint32_t* cnts = (const uint32_t*)dev_counts;
  // ~100 keys per bucket: a lookup is a binary search inside its bucket (7 steps instead of 10 at 640 per bucket), and
  // the adjacency, routing and seed kernels are made of lookups
  int bits = 0;
  const uint64_t per_bucket = 96;
  while (bits < 23 && (n >> bits) > per_bucket) bits++;        // (2^24 buckets x 256 threads would be 2^32 work-items: one too many for a dispatch)
  for (int attempt = 0; attempt < 4; attempt++) {
    int b1 = (bits + 1) / 2;
    if (bits - b1 > 15) b1 = bits - 15;
    int b2 = bits - b1;
    int nb1 = 1 << b1;
    void *p, *pa, *pb, *pc, *pca, *pcb;
    int rc;
    if ((rc = g_ws[0].get((size_t)nb1 * 16 + 64, &p))) return rc;
    if ((rc = g_ws[1].get((n + 2) * 8, &pa))) return rc;
    if ((rc = g_ws[2].get((n + 2) * 8, &pb))) return rc;
    if ((rc = g_ws[3].get((n + 2) * 4, &pc))) return rc;
    if ((rc = g_ws[5].get((n + 2) * 4, &pca))) return rc;
    if ((rc = g_ws[6].get((n + 2) * 4, &pcb))) return rc;
    uint32_t* d_h = (uint32_t*)p;               // [nb1] hist, [nb1] off, [nb1] cursor, then seg[2] (u64) + total
    uint32_t* d_o = d_h + nb1;
    uint32_t* d_c = d_o + nb1;
    uint64_t* d_seg = (uint64_t*)(d_c + nb1 + (nb1 & 1) + 2);
    unsigned long long* d_tot = (unsigned long long*)(d_seg + 2);
    uint64_t seg[2] = {0, n};
    HIP_TRY(hipMemsetAsync(d_h, 0, (size_t)nb1 * 4, s));
    HIP_TRY(hipMemsetAsync(d_tot, 0, 8, s));
    HIP_TRY(hipMemcpyAsync(d_seg, seg, 16, hipMemcpyHostToDevice, s));
    uint32_t tiles = (uint32_t)std::max<uint64_t>(1, cdiv(n, TILE_KEYS));
    {
      TimerRegion t(ctx, T_HIST1);
      hipLaunchKernelGGL(hist_keys_kernel, dim3(tiles, 1), dim3(BLK), nb1 * 4, s, keys, d_seg, 1u, bits, b2, (uint32_t)nb1, d_h, (uint32_t)TILE_KEYS);
      hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(BLK), 0, s, d_h, nb1, d_o, d_c);
      hipLaunchKernelGGL(sum_counts_kernel, dim3(256), dim3(256), 0, s, cnts, n, d_tot);
    }
    std::vector<uint32_t> h(nb1);
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(h.data(), d_h, (size_t)nb1 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&total, d_tot, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<uint64_t> off1(nb1 + 1);
    uint64_t a = 0;
    for (int i = 0; i < nb1; i++) { off1[i] = a; a += h[i]; }
    off1[nb1] = a;
    {
      TimerRegion t(ctx, T_SCATTER1);
      hipLaunchKernelGGL((scatter_keys_kernel<true>), dim3(tiles, 1), dim3(BLK), nb1 * 8, s, keys, cnts, d_seg, 1u, bits, b2, (uint32_t)nb1, d_c,
                         (uint64_t*)pa, (uint32_t*)pca, (uint32_t)TILE_KEYS);
    }
    bool ov = false;
    rc = build_from_keys(ctx, (uint64_t*)pa, (uint64_t*)pb, (uint32_t*)pc, (uint32_t*)pca, (uint32_t*)pcb, off1, bits, b2, k1,
                         canonical, n, total, out, &ov);
    if (rc) return rc;
    if (!ov) return SHN_OK;
    shn_table_destroy(*out);
    *out = nullptr;
    bits = std::min(23, bits + 2);
  }
  return shn_fail(SHN_ERR_OVERFLOW, "shn_table_from_pairs: bucket overflow after 4 attempts");
}

#define SHARD_SALT 0xA24BAED4963EE407ULL
__device__ __forceinline__ uint32_t owner_of(uint64_t key, int n_ranks) {
  return (uint32_t)(shn_mix64(key ^ SHARD_SALT) % (uint64_t)n_ranks);
}
// mode 0: by the key's hash; mode 1: by its minimizer (k, canon: of the table)
struct ShardRule { int mode, k, canon; };
__device__ __forceinline__ uint32_t owner_by(const ShardRule& R, uint64_t key, int n_ranks) {
  return R.mode ? shn_owner_minimizer(key, R.k, R.canon, n_ranks) : owner_of(key, n_ranks);
}
// Two passes over the keys with the SAME launch shape: the first leaves, per block and rank, how many of the block's keys the rank
// owns; an exclusive scan of those counts (rank-major) is where every block writes every rank's keys in the second pass -- inside the
// block an LDS cursor per rank.  No global atomics: with one cursor per rank in HBM (wave-aggregated, until round 5) a shard pass
// over 183 M keys took 118 ms, bound by the atomics on those few addresses.
#define SHARD_GRID 2048
__global__ void shard_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int n_ranks, const ShardRule R, unsigned long long* __restrict__ hist,
                                  uint32_t* __restrict__ block_count) {
  __shared__ uint32_t lh[64];
  if (threadIdx.x < 64) lh[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&lh[owner_by(R, keys[i], n_ranks)], 1u);
  __syncthreads();
  if (threadIdx.x < n_ranks) {
    block_count[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = lh[threadIdx.x];
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
  }
}
__global__ void shard_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, uint64_t n, int n_ranks,
                                     const ShardRule R, const uint64_t* __restrict__ pos, uint64_t* __restrict__ ok, uint32_t* __restrict__ oc) {
  __shared__ uint32_t lcur[64];
  __shared__ uint64_t lbase[64];
  if (threadIdx.x < 64) { lcur[threadIdx.x] = 0; lbase[threadIdx.x] = (int)threadIdx.x < n_ranks ? pos[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] : 0; }
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];
    const uint32_t o = owner_by(R, key, n_ranks);
    const uint64_t d = lbase[o] + atomicAdd(&lcur[o], 1u);
    ok[d] = key; oc[d] = counts[i];
  }
}

extern "C" int shn_table_shard_mode(shn_ctx* ctx, const shn_table* t, int n_ranks, int mode, uint64_t* per_rank, void* dev_keys_out, void* dev_counts_out);
extern "C" int shn_table_shard(shn_ctx* ctx, const shn_table* t, int n_ranks, uint64_t* per_rank, void* dev_keys_out,
                               void* dev_counts_out) {
  return shn_table_shard_mode(ctx, t, n_ranks, 0, per_rank, dev_keys_out, dev_counts_out);
}
// mode 1: owner = rank of the key's minimizer (shn_owner_minimizer) -- the shards the components are labelled on (shn_cc_*)
extern "C" int shn_table_shard_mode(shn_ctx* ctx, const shn_table* t, int n_ranks, int mode, uint64_t* per_rank, void* dev_keys_out,
                                    void* dev_counts_out) {
  if (!ctx || !t || !per_rank || n_ranks < 1 || n_ranks > 64 || mode < 0 || mode > 1) return shn_fail(SHN_ERR_ARG, "shn_table_shard: bad argument");
  const ShardRule R{mode, t->k, t->canonical};
  if (t->n && (!dev_keys_out || !dev_counts_out)) return shn_fail(SHN_ERR_ARG, "shn_table_shard: NULL output");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, cdiv(t->n, 256)), SHARD_GRID);
  const uint64_t nbc = (uint64_t)n_ranks * grid;
  void* p;
  int rc;
  if ((rc = g_ws[0].get(64 * 8 + nbc * 4 + 64 + (nbc + 2) * 8, &p))) return rc;
  unsigned long long* d_hist = (unsigned long long*)p;
  uint64_t* d_pos = (uint64_t*)(d_hist + 64);
  uint32_t* d_bc = (uint32_t*)(d_pos + nbc + 2);
  HIP_TRY(hipMemsetAsync(d_hist, 0, 64 * 8, s));
  hipLaunchKernelGGL(shard_hist_kernel, dim3(grid), dim3(256), 0, s, t->d_keys, t->n, n_ranks, R, d_hist, d_bc);
  if ((rc = shn_device_scan_u32(ctx, d_bc, nbc, d_pos, nullptr))) return rc;
  unsigned long long h[64];
  HIP_TRY(hipMemcpyAsync(h, d_hist, 64 * 8, hipMemcpyDeviceToHost, s));
  hipLaunchKernelGGL(shard_scatter_kernel, dim3(grid), dim3(256), 0, s, t->d_keys, t->d_counts, t->n, n_ranks, R, d_pos,
                     (uint64_t*)dev_keys_out, (uint32_t*)dev_counts_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n_ranks; i++) per_rank[i] = h[i];
  return SHN_OK;
}
