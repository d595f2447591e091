// Shared host/device helpers for libshannon_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/shannon_hip.h"

// Copy census (SHN_COPY_CENSUS=1, a diagnostic): every hipMemcpyAsync / hipMemsetAsync of the library counted by call site and
// printed when the process ends -- which of the thousands of small copies of a step come from where (core.hip).
hipError_t shn_counted_memcpy(const char* file, int line, void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s);
hipError_t shn_counted_memset(const char* file, int line, void* dst, int value, size_t n, hipStream_t s);
#define hipMemcpyAsync(...) shn_counted_memcpy(__FILE__, __LINE__, __VA_ARGS__)
#define hipMemsetAsync(...) shn_counted_memset(__FILE__, __LINE__, __VA_ARGS__)

#define SHN_WAVE 64

void shn_set_error(const std::string& msg);
int shn_fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return shn_fail(SHN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
  } while (0)

enum {
  T_PACK = 0, T_HIST1, T_SCATTER1, T_HIST2, T_SCATTER2, T_COUNT, T_COMPACT, T_COUNT_TOTAL, T_LOOKUP,
  T_EXTEND, T_ROUTE, T_GRAPH, T_LP, T_EXT_PREP, T_EXT_SORT, T_EXT_WALK, T_SEEDS, T_EXT_WALK_THREAD, T_EXT_WALK_WAVE, T_EXT_MARK, T_EXT_EMIT, T_TABLE_BUILD, T_COUNT_DIRECT, T_CONTIG, T_GRAPH_GPU, T_EXT_ADJ,
  T_SK_HIST, T_SK_EMIT, T_SK_HIST2, T_SK_SCATTER2, T_SK_BUCKETS, T_SK_BIG, T_EXT_WALK_FRESH, T_EXT_BEGIN, T_SK_BUCKETS2,
  // round 6: the kernels of the contig stage, of the read -> graph mapping and the LP trials, one launch per region (bench.py's kernel table)
  T_CG_SORT, T_CG_HITS, T_CG_COVER, T_CG_COMPACT, T_KP_SEARCH, T_KP_CLASSIFY, T_SEED_SCAN, T_DD_INSERT, T_LP_TRIALS, T_EXT_AUDIT, T_N = 48
};

// grow-only device workspace slot (process-wide ones: g_shn_ws below; per-context ones: shn_ctx::cws)
struct ShnWsSet;
struct ShnWs {
  void* p = nullptr; size_t cap = 0; uint64_t stage = 0;
  ShnWsSet* set = nullptr;             // the stage-owned set the slot belongs to (nullptr: a context's own slot, shn_ctx::cws)
  int get(size_t bytes, void** out);
};
// The workspaces of the top-level stages (count, extension, contig stage, probe table, routing, unitigs): 32 grow-only slots that
// the stages of ONE pipeline use one after the other on one host thread.  The process has a default set; a context may own a set
// (shn_ctx_own_workspaces) so that a second pipeline -- the back half of bench.py's two batches in flight, cut behind the extension
// -- runs its routing beside the first one's extension.
struct ShnWsSet {
  ShnWs ws[32];
  uint64_t stage = 1;                  // (under the workspaces' mutex, count.hip)
  uint64_t stage_thread = 0;
  ShnWsSet() { for (auto& w : ws) w.set = this; }
};

// grow-only pinned host buffer (staging of many small pieces as ONE transfer)
struct ShnPinned {
  void* p = nullptr; size_t cap = 0;
  int get(size_t bytes, void** out) {
    if (bytes > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr; cap = 0;
      const size_t want = bytes + bytes / 2 + 4096;
      if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return shn_fail(SHN_ERR_NOMEM, "hipHostMalloc (staging buffer)"); }
      cap = want;
    }
    *out = p;
    return 0;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct shn_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev0[T_N], ev1[T_N];
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[T_N];
  double ms[T_N];
  uint64_t regions[T_N];
  uint64_t abytes[T_N];    // algorithmic bytes of the launches of a slot, said by the launch site (TimerRegion::bytes): shn_timer_bytes
  bool timing;
  bool owns_stream;        // shn_ctx_fork: the stream is destroyed with the context
  double sk_pool_ratio;    // super-k-mer counting: (key, count) pairs per window the buckets emitted last time on this context (0: none yet)
  int count_direct_log2;   // one-pass counting: table size that sufficed last time (0 none yet, -1 gave up), shn_count_k1mers
  ShnWs cws[12];           // per-context workspaces of the calls several host threads make at the same time, each on its own
                           // context / stream (the graph threads' seed scans): [0] scan block sums, [1] [2] seed-scan counts / offsets,
                           // [4..11] the LP batches of the sparse flow (two batches may be in flight on two contexts)
  ShnPinned hpin[2];       // pinned staging of the LP batches: [0] what goes up, [1] what comes back
  ShnWsSet* wsset = nullptr;   // the stage workspaces this context's top-level stages use: the process's default set, or its own
  bool owns_wsset = false;
  shn_ctx* parent = nullptr;   // shn_ctx_fork: the context this one was forked from (its timers and LP census are the parent's: core.hip)
  std::mutex tmu;          // guards pending / ms / regions when another thread drains them
  int lp_rule;             // SHN_LP_RULE_CENTER (default) / SHN_LP_RULE_VERTEX (SHN_LP_RULE=vertex in the environment, shn_lp_set_rule)
  uint64_t lp_stats[8];    // shn_lp_stats
};

// RAII-less region timer: records events on the ctx stream; durations are summed lazily.
struct TimerRegion {
  shn_ctx* c; int slot; hipEvent_t a, b;
  hipStream_t st;
  TimerRegion(shn_ctx* ctx, int s);
  TimerRegion(shn_ctx* ctx, int s, hipStream_t stream);     // a region on another stream (one kernel launch)
  ~TimerRegion();
  // the algorithmic bytes of this launch (what the kernel has to read and write at the least: the byte model stated next to the
  // launch), summed per slot and read by bench.py's kernel table beside the slot's time
  void bytes(uint64_t b) { if (c && c->timing) __atomic_fetch_add(&c->abytes[slot], b, __ATOMIC_RELAXED); }
};

struct TimingOff {
  shn_ctx* c; bool old;
  explicit TimingOff(shn_ctx* ctx);
  ~TimingOff();
};

struct shn_reads {
  shn_ctx* ctx;
  int device;
  uint64_t n_reads;
  uint32_t fixed_len;      // 0 => ragged
  uint32_t max_len;
  uint64_t total_bases;
  uint64_t n_invalid;      // reads with >=1 non-ACGT base
  uint32_t wpr;            // 64-bit words per read slot (fixed) -- ragged reads use woff
  uint64_t n_words;
  uint64_t* d_words;       // 2-bit bases, MSB-first, every read starts on a word boundary
  uint64_t* d_mask;        // 1 bit per base (MSB-first in the same word geometry: bit for base j of
                           // the read's word w is bit 63-(j%32)*... see pack kernel), NULL if no N
  uint64_t* d_woff;        // ragged: word offset of each read (n_reads+1), else NULL
  uint32_t* d_len;         // ragged: length of each read, else NULL
  uint8_t* d_bad;          // per read: 1 if it contains a non-ACGT base (NULL if none)
  bool cached;             // device arrays come from the caching allocator (shn_reads_gather), not hipMalloc
};

// text on the device (the candidate contigs as shn_ext_emit_device leaves them): n bytes + 64 zeroed
struct shn_devtext { shn_ctx* ctx; uint8_t* d; uint64_t n; };

struct shn_table {
  shn_ctx* ctx;
  int device;
  int k;
  int canonical;
  uint64_t n;              // distinct keys
  uint64_t total;          // windows counted
  int bits;                // number of hash bits indexing buckets
  uint64_t n_buckets;
  uint64_t* d_keys;        // [n] grouped by bucket, ascending inside a bucket
  uint32_t* d_counts;      // [n]
  uint64_t* d_bucket_off;  // [n_buckets+1]
  int layout;              // 0: buckets = top `bits` bits of fmix64(key) (every table made from pairs); 1: buckets = shn_minimizer_bucket
                           // (the tables of the super-k-mer counting path, count_sk.hip): k1-mers that overlap by k1 - 1 bases mostly share
                           // their minimizer, so a k1-mer and its 8 neighbours mostly lie in ONE bucket
  int sk_m;                // layout 1: length of the minimizer
};

// ---------------------------------------------------------------- device helpers
__host__ __device__ __forceinline__ uint64_t shn_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ uint64_t shn_revcomp(uint64_t key, int k) {
  uint64_t v = __brevll(~key);
  v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
  return v >> (64 - 2 * k);
}

__host__ inline uint64_t shn_revcomp_host(uint64_t key, int k) {
  uint64_t out = 0;
  for (int i = 0; i < k; i++) { out = (out << 2) | (3 - (key & 3)); key >>= 2; }
  return out;
}

// 2k-bit window starting at base `pos` of a read whose words start at `w` (MSB-first packing).
__device__ __forceinline__ uint64_t shn_extract(const uint64_t* __restrict__ w, uint32_t pos, int k) {
  uint32_t wi = pos >> 5, sh = (pos & 31) * 2;
  uint64_t hi = w[wi];
  uint64_t v = hi << sh;
  if (sh + 2 * k > 64) v |= w[wi + 1] >> (64 - sh);   // sh > 0 here
  return v >> (64 - 2 * k);
}

// k mask bits (1 = non-ACGT) starting at base `pos`; mask words hold 64 bases each, MSB-first.
__device__ __forceinline__ uint64_t shn_extract_mask(const uint64_t* __restrict__ m, uint32_t pos, int k) {
  uint32_t wi = pos >> 6, sh = pos & 63;
  uint64_t v = m[wi] << sh;
  if (sh + k > 64) v |= m[wi + 1] >> (64 - sh);
  return v >> (64 - k);
}

__device__ __forceinline__ uint32_t shn_bucket_of(uint64_t key, int bits) {
  return bits ? (uint32_t)(shn_mix64(key) >> (64 - bits)) : 0u;
}

// ---- minimizers (count_sk.hip makes the tables of layout 1 with them; their consumers find a key's bucket with them)
// order of the m-mers: a bijection of their 2m <= 32 bits (murmur3's finaliser), so two different m-mers never tie
__host__ __device__ __forceinline__ uint32_t shn_sk_order(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
// the bucket of a minimizer: a second mix of its order value (minimizers are the SMALL order values: their top bits are no use)
__host__ __device__ __forceinline__ uint32_t shn_sk_bucket(uint32_t ord, int bits) {
  uint32_t x = ord * 0x9E3779B1u;
  x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
  return bits ? x >> (32 - bits) : 0u;
}
__device__ __forceinline__ uint32_t shn_revcomp32(uint32_t v, int m) {
  uint32_t x = __brev(~v);
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  return x >> (32 - 2 * m);
}
// smallest order value among the k - m + 1 m-mers of a k-mer (canon: of their canonical forms -- the same for both strands)
__device__ __forceinline__ uint32_t shn_minimizer_order(uint64_t key, int k, int m, int canon) {
  const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
  uint32_t best = 0xFFFFFFFFu;
  for (int i = 0; i <= k - m; i++) {
    const uint32_t f = (uint32_t)(key >> (2 * (k - m - i))) & mmask;
    uint32_t c = f;
    if (canon) { const uint32_t r = shn_revcomp32(f, m); c = r < f ? r : f; }
    const uint32_t o = shn_sk_order(c);
    best = o < best ? o : best;
  }
  return best;
}
__device__ __forceinline__ uint32_t shn_minimizer_bucket(uint64_t key, int k, int m, int canon, int bits) {
  return shn_sk_bucket(shn_minimizer_order(key, k, m, canon), bits);
}

// owner rank of a k1-mer by its minimizer (the N-rank path that labels components on the owner shards: a k1-mer and its eight
// neighbours share their minimizer six times out of seven, so most edges of the k1-mer graph stay inside a shard)
#define SHN_OWNER_M 13
__device__ __forceinline__ uint32_t shn_owner_of_order(uint32_t order, int world) {
  return (uint32_t)(shn_mix64((uint64_t)order ^ 0x6A09E667F3BCC909ULL) % (uint64_t)world);
}
__device__ __forceinline__ uint32_t shn_owner_minimizer(uint64_t key, int k, int canon, int world) {
  const int m = k < SHN_OWNER_M ? k : SHN_OWNER_M;
  return shn_owner_of_order(shn_minimizer_order(key, k, m, canon), world);
}

// A table as its consumers see it: keys grouped by bucket, ascending inside a bucket; the bucket of a key by the table's layout.
struct TabIdx {
  const uint64_t* keys; const uint64_t* boff; int bits, layout, k, m, canon;
};
static inline TabIdx shn_tab_idx(const shn_table* t) {
  TabIdx T; T.keys = t->d_keys; T.boff = t->d_bucket_off; T.bits = t->bits; T.layout = t->layout; T.k = t->k; T.m = t->sk_m; T.canon = t->canonical;
  return T;
}
__device__ __forceinline__ uint32_t shn_tab_bucket(const TabIdx& T, uint64_t key);
__device__ __forceinline__ int64_t shn_tab_find(const TabIdx& T, uint64_t key);

// index of `key` in a table (grouped by bucket, ascending inside a bucket) or -1
__device__ __forceinline__ int64_t shn_table_find(const uint64_t* __restrict__ tkeys, const uint64_t* __restrict__ boff,
                                                  int bits, uint64_t key) {
  uint32_t b = shn_bucket_of(key, bits);
  uint64_t lo = boff[b], hi = boff[b + 1];
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    uint64_t v = tkeys[mid];
    if (v == key) return (int64_t)mid;
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return -1;
}

__device__ __forceinline__ uint32_t shn_tab_bucket(const TabIdx& T, uint64_t key) {
  return T.layout ? shn_minimizer_bucket(key, T.k, T.m, T.canon, T.bits) : shn_bucket_of(key, T.bits);
}
__device__ __forceinline__ int64_t shn_tab_find(const TabIdx& T, uint64_t key) {
  const uint32_t b = shn_tab_bucket(T, key);
  uint64_t lo = T.boff[b], hi = T.boff[b + 1];
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const uint64_t v = T.keys[mid];
    if (v == key) return (int64_t)mid;
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return -1;
}

// The same for callers that know the key width (2k bits).  The keys of a bucket are a uniform sample of the key range (the bucket
// is picked by a hash of the key, the order inside is by key), so the key's top bits give its rank within +-sqrt(n): the search
// starts there and gallops outwards.  A bucket of ~90 keys spans 11 64-byte sectors; bisection touches 4-5 of them, this 1-2.
// Measured at BASELINE configs[2]: route_kernel (mostly hits) 126 -> 102 ms; ext_adjacency_kernel (mostly misses, 16 look-ups per
// k1-mer side by side in a wavefront) 540 -> 610 ms -- it keeps the bisection.
__device__ __forceinline__ int64_t shn_table_find_k(const uint64_t* __restrict__ tkeys, const uint64_t* __restrict__ boff,
                                                    int bits, uint64_t key, int kbits) {
  const uint32_t b = shn_bucket_of(key, bits);
  uint64_t lo = boff[b], hi = boff[b + 1];
  if (lo >= hi) return -1;
  const uint64_t n = hi - lo;
  const uint64_t f = kbits >= 16 ? (key >> (kbits - 16)) & 0xFFFFULL : (key << (16 - kbits)) & 0xFFFFULL;
  uint64_t g = lo + ((f * n) >> 16);
  if (g >= hi) g = hi - 1;
  uint64_t v = tkeys[g];
  if (v == key) return (int64_t)g;
  if (v < key) {
    lo = g + 1;
    for (uint64_t step = 1; lo < hi; step <<= 1) {
      const uint64_t r = g + step;
      if (r >= hi) break;
      v = tkeys[r];
      if (v == key) return (int64_t)r;
      if (v > key) { hi = r; break; }
      lo = r + 1;
    }
  } else {
    hi = g;
    for (uint64_t step = 1; lo < hi; step <<= 1) {
      if (g < lo + step) break;
      const uint64_t l = g - step;
      v = tkeys[l];
      if (v == key) return (int64_t)l;
      if (v < key) { lo = l + 1; break; }
      hi = l;
    }
  }
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    v = tkeys[mid];
    if (v == key) return (int64_t)mid;
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return -1;
}

// grow-only device workspace slots shared by the translation units (one process per GPU)
// Caching device allocator for the per-call objects (tables, extension state, routes): a freed block is kept and
// handed out again to the next request it fits (hipMalloc/hipFree of hundreds of MB cost milliseconds per step).
hipError_t shn_dev_malloc_raw(void** p, size_t bytes);                 // on the calling thread's current stream
hipError_t shn_dev_malloc_on(void** p, size_t bytes, hipStream_t stream);
void shn_dev_free(void* p);                  // (the block stays ordered behind its own stream and the calling thread's current stream: core.hip)
void shn_dev_free_on(void* p, hipStream_t stream);
void shn_dev_trim();                         // give the cached blocks back to the driver
// hipMalloc for the objects that outlive a call (read sets): when the driver says no, what the caching allocator keeps for the next
// call and the workspaces of earlier stages are given back first
hipError_t shn_hip_malloc(void** p, size_t bytes);
template <class T> hipError_t shn_hip_malloc(T** p, size_t bytes) { return shn_hip_malloc((void**)p, bytes); }
template <class T> static inline hipError_t shn_dev_malloc(T** p, size_t bytes) { return shn_dev_malloc_raw((void**)p, bytes); }
void shn_stream_retired(hipStream_t s);      // a stream about to be destroyed (synchronised by the caller): no block waits on it any more
void shn_use_stream(hipStream_t s);          // the calling host thread works on this stream from now on (what the allocator orders frees against)
hipStream_t shn_current_stream();
void shn_poison(void* p, size_t bytes, hipStream_t s);      // SHN_DEV_POISON: fill with the poison byte (else nothing)
void shn_debug_count(int i);                 // shn_debug_counter(i)++: [0] double frees [1] foreign frees [2] workspace slots asked for by a thread that does not own the stage [3] hand-overs without an event [4] cross-stream hand-overs
// entry of a call on a context: device + the thread's current stream
#define SHN_ENTER(ctx) do { HIP_TRY(hipSetDevice((ctx)->device)); shn_use_stream((ctx)->stream); } while (0)

void shn_stage_begin(shn_ctx* ctx);          // a top-level GPU stage starts on the context's workspace set (slots used before it become reclaimable)
ShnWsSet* shn_default_wsset();               // the process's set (count.hip)
ShnWsSet* shn_wsset_create();                // a set of its own for a context (registered with the reclaiming pass)
void shn_wsset_destroy(ShnWsSet* s);
static inline ShnWs* shn_ws(shn_ctx* ctx) { return (ctx->wsset ? ctx->wsset : shn_default_wsset())->ws; }
size_t shn_ws_release_idle();                // frees the slots not used by the current stage; returns the bytes given back
extern "C" int shn_host_cpus(void);
void shn_lp_census_add(shn_ctx* c, const uint64_t* v8);   // core.hip: LP census of a context or, for a fork, of its parent (under the forks' lock)
shn_ctx* shn_thread_ctx(shn_ctx* parent);      // core.hip: the calling host thread's own fork (stream) of a context
int shn_device_scan_u32(shn_ctx* ctx, const uint32_t* d_in, uint64_t n, uint64_t* d_out /* n+1 */, uint64_t* total_host);
// stable LSD radix sort of (u64 key, u32 value) pairs on bits [bit_lo, bit_hi); result lands in keys/vals
int shn_sort_pairs(shn_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, uint64_t n,
                   int bit_lo, int bit_hi);

// the same for 64-bit words alone (a value packed under the key); *sorted = keys or keys_tmp, whichever holds the result
int shn_sort_keys(shn_ctx* ctx, uint64_t* keys, uint64_t* keys_tmp, uint64_t n, int bit_lo, int bit_hi, uint64_t** sorted);

static inline uint64_t cdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// device buffers of one call on one stream, given back to the caching allocator when the call ends (the allocator orders the
// next use of a block behind what is still queued on this stream: an early error return needs no synchronisation of its own)
struct ShnDevBufs {
  std::vector<void*> p;
  hipStream_t stream;
  explicit ShnDevBufs(hipStream_t s) : stream(s) {}
  template <class T> hipError_t get(T** out, size_t bytes) { hipError_t e = shn_dev_malloc_on((void**)out, bytes, stream); if (e == hipSuccess) p.push_back((void*)*out); return e; }
  ~ShnDevBufs() { for (void* q : p) shn_dev_free_on(q, stream); }
};
// contig texts on the device (csrc/contig_gpu.hip): cid[g] = contig of base g; the k-windows of the selected contigs (use ==
// NULL: all) as (packed key, base index of the window start) pairs sorted by key, stable (so by contig, position inside a run)
void shn_contig_ids(hipStream_t s, const uint64_t* d_off, uint64_t n_contigs, uint32_t* d_cid);
int shn_sorted_windows(shn_ctx* ctx, ShnDevBufs& bufs, const uint8_t* d_bases, const uint64_t* d_off, const uint32_t* d_cid, const int32_t* d_use,
                       uint64_t total, int k, uint64_t** keys, uint32_t** vals, uint64_t* n_out);
