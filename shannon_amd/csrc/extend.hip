// Contig extension / k1-mer error correction on gfx950 (rows a3-a4): replaces the sequential
// heaviest-first greedy walk of extension_correction.py:334-354 (load_kmers :202-221, extend
// :223-245, argmax :159-166).
//
// The reference processes seeds one by one in (weight desc, k1-mer asc) order with a global
// `traversed` set.  Here every seed is a walk with priority = its rank in that order and all
// walks run in parallel as a fixpoint iteration:  walk r treats a k1-mer as traversed iff it was
// claimed (previous iteration) by a walk of smaller rank, or is on its own trail.  The sequential
// result is the unique fixpoint (walk 0 is right after one iteration, walk r once every walk < r
// it touches is right), so iterating until no walk changes reproduces the reference exactly.
//
// The iteration is change-driven (a worklist): claims persist in one array; after a round every k1-mer whose
// owner changed marks as dirty the walks whose view it changes -- the walks standing next to it (owners of its 8
// neighbours) and the walk seeded on it, and only those for which it BECAME available (old owner < walk < new
// owner).  Only dirty walks run in the next round, after their old claims have been released; the round count
// stays the dependency depth of the data, but a round costs the affected walks, not all of them.  A consistent
// state (no dirty walk) is the unique fixpoint = the sequential result.
// Memos: after every round the path of each walk that ran alive is rebuilt from the claims into a memo slot, and
// every k1-mer remembers the (walk, step) it was written under (hint).  A wavefront re-checks 64 memo steps per
// memory round trip -- of its own memo, or of the memo of whatever walk last owned the k1-mer it reached, in either
// direction -- and walks sequentially only in between.  Memos are hints: every entry is validated against the
// hint of its k1-mer, and a changed decision is re-made against the live claims.
// Short walks run one per thread; one that turns out long hands over to a wavefront in the same round.
// shn_extend_sharded: walks never leave their connected component of the k1-mer graph, so the components (GPU
// union-find over the adjacency rows) can be dealt to several ranks.
//
// Oriented k1-mers: the count table stores canonical keys; oriented id o = 2*i + s is the string
// key_i (s=0) or its reverse complement (s=1; unused for palindromes).  Both strands are walked,
// as in the reference's strand-doubled input.
#include "common.h"
#include <cstring>
#include <vector>
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define EBLK 256
// the thread walkers run in workgroups of ONE wavefront: a workgroup keeps its place on the CU (8 per CU at 256 threads) until its
// last wavefront has ended, and a walk kernel's wavefronts end at very different times (each runs as long as its longest walk) --
// measured with 256-thread workgroups: 1,750 of 8,192 wavefront slots in use on average over a bulk round
#define WBLK 64
#define UNCLAIMED 0xFFFFFFFFu
#define NONE32 0xFFFFFFFFu
#define UNCLAIMED64 0xFFFFFFFFFFFFFFFFULL
#define NOHINT_WORD 0xFFFFFFFFu   // = NOHINT (defined with the memo machinery below)
#define LONG_WALK 8           // dirty walks at least this long (last run or memo) get a wavefront
#define MEMO_MIN 1            // walks at least this long get a memo slot
#define PROMOTE_STEPS 16      // a thread walker that gets this far hands over to a wavefront
typedef unsigned long long u64;
#define CLAIM(rank, pos) (((u64)(rank) << 32) | (u64)(uint32_t)(pos))
#define RANK(c) ((uint32_t)((c) >> 32))
#define POS(c) ((uint32_t)(c))

struct Adj4 { int32_t v[4]; };
// Everything about an oriented k1-mer that does not change while the walks iterate, in ONE 64-byte line: both adjacency rows, its
// weight, and the two words the mark pass keeps per k1-mer (memo hint, rank of the walk seeded on it).  A walk step used to touch
// four arrays per candidate (claims, snapshot, weights, rows); the rows and the weight of a candidate -- and whatever the mark
// pass needs around a changed k1-mer -- now come with one sector.  The bulk rounds are bound by the NUMBER of random 64-byte
// sectors the chip serves (~30 G/s measured, at any lane occupancy), so sectors per step is what the layout is chosen for.
// The claims and their snapshot stay compact arrays of their own: the begin / mark / audit / emit passes stream them.
struct __attribute__((aligned(64))) Rec {
  Adj4 R;                // oriented id reached by appending base b, or -1
  Adj4 L;                // ... by prepending base b
  uint32_t weight;       // weight of the string in the doubled input
  uint32_t hint;         // where this k1-mer was last written into a memo (pool index << 2 | kind), NOHINT if never
  uint32_t seed_rank;    // rank of the walk seeded on it, 0xFFFFFFFF if it is not a seed
  uint32_t pad[5];
};
static_assert(sizeof(Rec) == 64, "one line per oriented k1-mer");
// rows of one direction / the weights / the hints, indexed by oriented id (strided views of the record array)
struct RowView { const char* p; __device__ __forceinline__ Adj4 operator[](uint32_t i) const { return *(const Adj4*)(p + ((uint64_t)i << 6)); } };
struct WordView { const char* p; __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return *(const uint32_t*)(p + ((uint64_t)i << 6)); } };
__host__ __device__ __forceinline__ RowView rows_R(const Rec* r) { return RowView{(const char*)r}; }
__host__ __device__ __forceinline__ RowView rows_L(const Rec* r) { return RowView{(const char*)r + 16}; }
__host__ __device__ __forceinline__ WordView words_weight(const Rec* r) { return WordView{(const char*)r + 32}; }
__host__ __device__ __forceinline__ WordView words_hint(const Rec* r) { return WordView{(const char*)r + 36}; }

// ---- stage checksums (SHN_EXT_DIGEST=1; tests/test_stress_gpu.py, tools/stress_digest.py): two runs on the same input must agree
// stage by stage; the first stage that differs -- and the 1/64 of its array where -- localises a run-to-run difference.
// Stages: 0 table keys, 1 table counts, 2 bucket offsets, 3 weights + flags, 4 records (adjacency rows, weight, seed rank; before the
// first round), 5 seed order, 6 converged claims, 7 walk records (n_right, n_left, total weight).
#define EXT_DIG_STAGES 8
#define EXT_DIG_CHUNKS 64
__global__ void ext_digest_kernel(const uint32_t* __restrict__ w, uint64_t n_words, uint64_t salt, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  uint32_t cur = 0xFFFFFFFFu;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t ch = (uint32_t)((i * EXT_DIG_CHUNKS) / n_words);
    if (ch != cur) { if (cur != 0xFFFFFFFFu && acc) atomicAdd(&out[cur], acc); cur = ch; acc = 0; }
    acc += shn_mix64((i * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)w[i] << 1) ^ salt);
  }
  if (cur != 0xFFFFFFFFu && acc) atomicAdd(&out[cur], acc);
}

struct shn_ext {
  shn_ctx* ctx;
  int device;
  int k;
  uint64_t n;            // canonical entries
  uint64_t n_seeds;
  int iterations;
  uint32_t min_weight;
  const shn_table* table;
  shn_table* owned_table; // sharded: the k1-mers of this rank's components (table points at it)
  uint32_t* d_weight;    // [n] weight of the string in the doubled input (count, x2 for palindromes)
  uint8_t* d_flags;      // [n] bit0 palindrome, bit1 low complexity
  Rec* d_rec;            // [2n] per oriented k1-mer: adjacency rows, weight, memo hint, seed rank (see Rec)
  uint32_t* d_order;     // [n_seeds] oriented id of the seed with rank r
  u64* d_claim;          // [2n] converged claims: (rank of the owning walk) << 32 | (1 + step index on its path)
  u64* d_claim2;         // [2n] scratch (the second half of the block d_claim starts: freed with it)
  uint64_t total_steps;  // walk steps executed over all iterations (for the bench's byte model)
  uint64_t wave_steps;   // ... of which by the wavefront kernel
  uint64_t fresh_steps;  // ... of which by the thread walker in the first round of a rank block
  int dense_rounds;      // rounds whose begin / mark passes streamed all claims
  uint64_t settled_walks; // walks that were never launched: a lower rank on a forced chain of their seed (ext_chain_has_lower)
  uint32_t* d_nr;        // [n_seeds] right steps (UNCLAIMED = void walk)
  uint32_t* d_nl;        // [n_seeds]
  uint64_t* d_totw;      // [n_seeds] sum of weights incl. the seed
  uint64_t dig[EXT_DIG_STAGES][EXT_DIG_CHUNKS];   // SHN_EXT_DIGEST=1: checksums of the stages' arrays (shn_ext_digests)
  int has_dig;
};

__device__ __forceinline__ uint64_t oriented_string(const uint64_t* __restrict__ tkeys, uint32_t o, int k) {
  uint64_t key = tkeys[o >> 1];
  return (o & 1) ? shn_revcomp(key, k) : key;
}

__global__ void ext_prepare_kernel(const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ tcounts, uint64_t n, int k,
                                   int canonical, uint32_t* __restrict__ weight, uint8_t* __restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = tkeys[i];
  uint64_t c = tcounts[i];
  uint8_t f = 0;
  if (canonical && shn_revcomp(key, k) == key) { f |= 1; c *= 2; }
  // lowComplexity (extension_correction.py:142-149): the most frequent base occurs >= k-2 times
  uint64_t lanes = (k == 32) ? 0x5555555555555555ULL : ((1ULL << (2 * k)) - 1) & 0x5555555555555555ULL;
  int mx = 0;
  for (uint64_t v = 0; v < 4; v++) {
    uint64_t pat = v * 0x5555555555555555ULL;
    uint64_t t = ~(key ^ pat);
    int cnt = __popcll((t & (t >> 1)) & lanes);
    mx = cnt > mx ? cnt : mx;
  }
  if (mx >= k - 2) f |= 2;
  weight[i] = (uint32_t)(c > 0xFFFFFFFFULL ? 0xFFFFFFFFULL : c);
  flags[i] = f;
}

// ---- the dictionary of the adjacency build: one 128-byte line per bucket -- ten keys (80 bytes), their ten id words (40 bytes),
// the number of keys that hashed here (4 bytes) -- so that a look-up, hit or miss, is ONE fetch.  HBM serves 128 bytes per request
// whatever is asked for (profiles/r03_fetch_calibration.txt), and the build makes 5.8 G look-ups at configs[2] of which 70 % miss:
// through the count table (bucket offsets -> bisection of a ~90-key bucket) a look-up was ~5 fetches, with a Bloom filter and a
// separator record per bucket in front (round 2) 1 for most misses and 3 for a hit -- 1.5 TB per launch, the kernel sat at the
// HBM limit.  Five keys per line on average: a bucket overflows with probability 1.4 %; what does not fit is found through the
// count table (the line's count says that there is more).  id word = table index | palindrome << 31; low-complexity k1-mers are
// not entered (load_kmers drops them, extension_correction.py:202-221).  An empty slot holds key 0 = AAA...A, which is
// low-complexity and therefore never a valid answer.
#define FD_SLOTS 10
#define FD_PER_LINE 4         // keys per line on average: a line overflows with probability 0.3 % (5: 1.4 %)
#define FD_HOPS 4             // what does not fit its line goes into the next ones
#define FD_PAL 0x80000000u
// (the hash of the count table's buckets, so that the build -- which goes through the table in bucket order -- fills the lines
// front to back: its atomics stay in the L2 and the lines stream out once; with a hash of its own the build was 724 M random
// read-modify-writes, 70 ms)
// (tables of layout 1 -- buckets of minimizers: a bucket's keys spread over the bucket's own stretch of lines, so the build still
// streams, and since a k1-mer's eight neighbours mostly share its minimizer, their look-ups mostly fall into the stretch the block
// is working through -- lines the L2 already holds)
__device__ __forceinline__ uint64_t fd_bucket(const TabIdx& T, uint64_t key, uint64_t n_lines) {
  const uint64_t h = shn_mix64(key);
  if (!T.layout) return __umul64hi(h, n_lines);
  // (buckets of minimizers differ in size by orders of magnitude: a bucket's lines are its share of the table -- one line per
  // FD_PER_LINE keys, from where its keys begin -- and the key picks one of them)
  const uint32_t b = shn_tab_bucket(T, key);
  const uint64_t lo = T.boff[b], hi = T.boff[b + 1];
  return lo / FD_PER_LINE + b + __umul64hi(h, (hi - lo) / FD_PER_LINE + 1);
}
// the same with the key's bucket known (layout 1)
__device__ __forceinline__ uint64_t fd_line_in_bucket(const TabIdx& T, uint32_t b, uint64_t key) {
  const uint64_t lo = T.boff[b], hi = T.boff[b + 1];
  return lo / FD_PER_LINE + b + __umul64hi(shn_mix64(key), (hi - lo) / FD_PER_LINE + 1);
}
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one; each XCD has its own 4 MB L2).  A kernel that
// goes through the table in order -- consecutive blocks on consecutive k1-mers, whose dictionary look-ups fall into the same few
// lines -- therefore spreads every stretch of lines over eight L2s.  With this block number instead of blockIdx.x the blocks that
// share an XCD are consecutive: each XCD works through one contiguous eighth of the launch.  (Speed only; any mapping is correct.)
// MEASURED at BASELINE configs[2] (round 5, two boxes-worth of A/B on one box): ext_records_kernel 110 -> 119-120 ms WITH the
// mapping -- the spread over all eight L2s (and the shared infinity cache behind them) serves the look-ups better than one L2 per
// stretch; SHN_XCD_MAP=1 switches it on, the default is the plain order.
__device__ __forceinline__ uint64_t xcd_block(int on) {
  const uint32_t g8 = gridDim.x & ~7u;
  if (!on || blockIdx.x >= g8) return blockIdx.x;
  return (uint64_t)(blockIdx.x & 7u) * (g8 >> 3) + (blockIdx.x >> 3);
}
__global__ void fd_build_kernel(const TabIdx T, const uint8_t* __restrict__ flags, uint64_t n,
                                unsigned long long* __restrict__ lines, uint64_t n_lines, int xcd) {
  const uint64_t* __restrict__ tkeys = T.keys;
  const uint64_t i = xcd_block(xcd) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t f = flags[i];
  if (f & 2) return;
  const uint64_t key = tkeys[i];
  // (a full line sends the entry on to the next one -- the count word tallies the attempts, so a look-up that finds more than
  // FD_SLOTS there goes on as well; the dictionary ends in FD_HOPS spare lines)
  unsigned long long* line = lines + fd_bucket(T, key, n_lines) * 16;
  for (int hop = 0; hop < FD_HOPS; hop++, line += 16) {
    const uint32_t slot = atomicAdd((uint32_t*)line + 30, 1u);
    if (slot < FD_SLOTS) { line[slot] = key; ((uint32_t*)line)[20 + slot] = (uint32_t)i | ((f & 1) ? FD_PAL : 0u); break; }
  }
}
// One look-up by the eight lanes g0 .. g0+7 of a wavefront (p = lane - g0; all eight pass the same key): lane p holds bytes
// 16 p .. 16 p + 15 of the line -- one coalesced 128-byte request.  Returns the id word or 0xFFFFFFFF, the same in all eight lanes.
__device__ __forceinline__ uint64_t shfl_u64(uint64_t x, int src) {
  return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(x >> 32), src, 64) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)x, src, 64);
}
__device__ __forceinline__ uint32_t fd_match(ulonglong2 v, const unsigned long long* __restrict__ lines, uint64_t line, uint64_t key, int p, int g0,
                                             const TabIdx& T, const uint8_t* __restrict__ flags) {
  for (int hop = 0;; hop++) {
    const bool ok = p < 5 && key != 0;
    const unsigned long long m0 = (__ballot(ok && v.x == key) >> g0) & 0xFFULL, m1 = (__ballot(ok && v.y == key) >> g0) & 0xFFULL;
    if (m0 | m1) {
      const int slot = m0 ? 2 * (__ffsll((long long)m0) - 1) : 2 * (__ffsll((long long)m1) - 1) + 1;
      // id words: bytes 80 .. 119 = words 20 .. 29: lane 5 + slot / 4, its word slot % 4
      const uint32_t w = (slot & 2) ? ((slot & 1) ? (uint32_t)(v.y >> 32) : (uint32_t)v.y) : ((slot & 1) ? (uint32_t)(v.x >> 32) : (uint32_t)v.x);
      return (uint32_t)__shfl((int)w, g0 + 5 + (slot >> 2), 64);
    }
    const uint32_t cnt = (uint32_t)__shfl((int)(uint32_t)(v.y >> 0), g0 + 7, 64) ;   // word 30 = low half of lane 7's second word
    if (cnt <= FD_SLOTS || key == 0) return 0xFFFFFFFFu;
    if (hop == FD_HOPS - 1) break;
    line++;                                                          // (the line overflowed, 0.3 % of them do: the next one -- one more fetch of the eight lanes)
    v = ((const ulonglong2*)(lines + line * 16))[p];
  }
  const int64_t j = shn_tab_find(T, key);                           // (FD_HOPS full lines in a row)
  if (j < 0) return 0xFFFFFFFFu;
  const uint8_t fj = flags[j];
  return (fj & 2) ? 0xFFFFFFFFu : ((uint32_t)j | ((fj & 1) ? FD_PAL : 0u));
}

// The records of both orientations of every canonical k1-mer from 8 look-ups instead of 16: the right candidates of the reverse-
// complement orientation are the reverse complements of the forward orientation's left candidates (rc(s)[1:] + b = rc(comp(b) +
// s[:-1])) and vice versa -- the same table entry j, the other orientation (the same one if entry j is its own reverse
// complement).  Eight lanes per canonical k1-mer: they make its eight look-ups together, one after the other (every look-up one
// 128-byte request of the eight lanes, all eight requests in flight before the first is looked at), and then write the two
// records -- 128 contiguous bytes -- 16 bytes each.
struct __attribute__((aligned(16))) Quad { uint32_t a, b, c, d; };
#ifndef SHN_REC_PIPE
#define SHN_REC_PIPE 0      // 1: the next trip's key / flag / weight fetched at the top of the trip.  Measured (round 5, configs[2]): 128 ms against 119.5 -- the
#endif                      // eight look-ups of a trip are already all in flight together and the extra live registers cost a wavefront or spills; left off

#if SHN_REC_PIPE
#define REC_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))      // (the pipelined form needs 105 registers: held at 96 = 5 wavefronts per SIMD, ten words spill)
#else
#define REC_KERNEL_ATTR
#endif
__global__ REC_KERNEL_ATTR void ext_records_kernel(const TabIdx T, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ weight, uint64_t n, int k, int canonical,
                                   Rec* __restrict__ rec, const unsigned long long* __restrict__ lines, uint64_t n_lines, int xcd) {
  const uint64_t* __restrict__ tkeys = T.keys;
  const uint64_t total = n * 8;
  const uint64_t rounded = (total + 63) & ~63ULL;                       // whole wavefronts take part in the ballots and shuffles
  const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const int lane = threadIdx.x & 63, g0 = lane & ~7, p = lane & 7;
  // (software-pipelined: the k1-mer, its flags and its weight of the NEXT trip are asked for at the top of this one -- a k1-mer is a
  // chain of dependent round trips (its key -> its neighbours' bucket offsets -> their dictionary lines), the kernel runs at 5
  // wavefronts per SIMD, and the first link of the chain need not be one)
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t gid = xcd_block(xcd) * blockDim.x + threadIdx.x;
  bool in = gid < total;
  uint64_t i = in ? gid >> 3 : 0;
  uint8_t f = (gid < rounded && in) ? flags[i] : (uint8_t)2;
  uint64_t str = gid < rounded ? tkeys[i] : 0ULL;
  uint32_t wt = (gid < rounded && in && (p == 2 || p == 6)) ? weight[i] : 0u;
  for (; gid < rounded;) {
    const uint64_t gid_n = gid + stride;
    const bool in_n = gid_n < total;
    const uint64_t i_n = in_n ? gid_n >> 3 : 0;
    uint8_t f_n = 2; uint64_t str_n = 0; uint32_t wt_n = 0;
    if (SHN_REC_PIPE && gid_n < rounded) { str_n = tkeys[i_n]; if (in_n) { f_n = flags[i_n]; if (p == 2 || p == 6) wt_n = weight[i_n]; } }
    const bool dead0 = (f & 2) != 0;                                    // forward orientation
    const bool dead1 = dead0 || (f & 1) || !canonical;                  // reverse-complement orientation (absent for palindromes)
    // look-up q: q = 0..3 append base q, q = 4..7 prepend base q - 4; lane q prepares it (key, strand, line), the group shares
    uint64_t mykey; uint32_t mystrand = 0;
    {
      const uint64_t b = (uint64_t)(p & 3);
      mykey = p < 4 ? (((str << 2) | b) & mask) : ((str >> 2) | (b << (2 * (k - 1))));
      if (canonical) { const uint64_t rc = shn_revcomp(mykey, k); if (rc < mykey) { mykey = rc; mystrand = 1; } }
      if (dead0) mykey = 0;                                             // (no look-up is needed: key 0 is never found)
    }
    uint64_t myline;
    if (T.layout) {
      // The minimizers of the eight neighbours from the k1-mer's own m-mers: the four successors share its m-mers 1 .. w - 1, the
      // four predecessors its m-mers 0 .. w - 2, and each adds one m-mer of its own (its last / first).  The eight lanes split the
      // k1-mer's w m-mers between them (a minimizer is the same on both strands, so the stored orientation serves); a look-up made
      // from scratch costs w order values per neighbour, and the kernel was bound by exactly that arithmetic.
      const int m = T.m, w = k - m + 1;
      const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
      uint32_t smin = 0xFFFFFFFFu, pmin = 0xFFFFFFFFu;                  // over m-mers 1 .. w - 1 / 0 .. w - 2
      for (int pos = p; pos < w; pos += 8) {
        const uint32_t f = (uint32_t)(str >> (2 * (k - m - pos))) & mmask;
        uint32_t c = f;
        if (canonical) { const uint32_t r = shn_revcomp32(f, m); c = r < f ? r : f; }
        const uint32_t o = shn_sk_order(c);
        if (pos >= 1) smin = o < smin ? o : smin;
        if (pos <= w - 2) pmin = o < pmin ? o : pmin;
      }
#pragma unroll
      for (int d = 1; d < 8; d <<= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)smin, d, 64), b2 = (uint32_t)__shfl_xor((int)pmin, d, 64);
        smin = a < smin ? a : smin; pmin = b2 < pmin ? b2 : pmin;
      }
      const uint32_t nb = (uint32_t)(p & 3);
      const uint32_t f = p < 4 ? ((((uint32_t)str & (mmask >> 2)) << 2) | nb)                          // the successor's last m-mer
                               : ((nb << (2 * (m - 1))) | ((uint32_t)(str >> (2 * (k - m + 1))) & (mmask >> 2)));   // the predecessor's first
      uint32_t c = f;
      if (canonical) { const uint32_t r = shn_revcomp32(f, m); c = r < f ? r : f; }
      uint32_t o = shn_sk_order(c);
      const uint32_t shared = p < 4 ? smin : pmin;
      o = shared < o ? shared : o;
      myline = fd_line_in_bucket(T, shn_sk_bucket(o, T.bits), mykey);
    } else myline = fd_bucket(T, mykey, n_lines);
    const uint32_t strands = (uint32_t)((__ballot(mystrand != 0) >> g0) & 0xFFULL);
    uint64_t key[8];
    ulonglong2 v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      key[q] = shfl_u64(mykey, g0 + q);
      v[q] = ((const ulonglong2*)(lines + shfl_u64(myline, g0 + q) * 16))[p];
    }
    uint32_t r8[8], d8[8];                                              // the candidate (oriented id), its other orientation
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t w = fd_match(v[q], lines, shfl_u64(myline, g0 + q), key[q], p, g0, T, flags);
      if (w == 0xFFFFFFFFu) { r8[q] = d8[q] = 0xFFFFFFFFu; continue; }
      const uint32_t j = w & ~FD_PAL;
      const uint32_t st = (strands >> q) & 1u;
      r8[q] = 2 * j + st;
      d8[q] = (w & FD_PAL) ? r8[q] : 2 * j + (1 - st);
    }
    if (in) {
      // lane p writes bytes 16 p .. of the pair of records (forward, reverse complement)
      Quad out;
      const Quad none = Quad{~0u, ~0u, ~0u, ~0u};
      switch (p) {
        case 0: out = Quad{r8[0], r8[1], r8[2], r8[3]}; break;           // forward: right row = the append candidates
        case 1: out = Quad{r8[4], r8[5], r8[6], r8[7]}; break;           //          left row = the prepend candidates
        case 4: out = dead1 ? none : Quad{d8[7], d8[6], d8[5], d8[4]}; break;   // reverse complement: right row = the prepend candidates' other orientations, mirrored (base b <-> 3 - b)
        case 5: out = dead1 ? none : Quad{d8[3], d8[2], d8[1], d8[0]}; break;   //                     left row = the append candidates' ...
        case 2: case 6: out = Quad{wt, NOHINT_WORD, 0xFFFFFFFFu, 0u}; break;
        default: out = Quad{0u, 0u, 0u, 0u}; break;
      }
      ((Quad*)(rec + 2 * i))[p] = out;
    }
    gid = gid_n; in = in_n; i = i_n;
    if (SHN_REC_PIPE) { f = f_n; str = str_n; wt = wt_n; }
    else if (gid < rounded) { str = tkeys[i]; f = in ? flags[i] : (uint8_t)2; wt = (in && (p == 2 || p == 6)) ? weight[i] : 0u; }
  }
}

// The same records for a table of layout 1 with most look-ups answered from LDS (round 5).  A k1-mer's neighbour has the k1-mer's
// own minimizer seven times out of eight, i.e. lies in the k1-mer's own bucket -- and the kernel above already knows every
// neighbour's bucket (it needs it for the dictionary line).  A block takes REC_G consecutive buckets (~170 keys each at BASELINE
// configs[2]), stages their keys and flags in LDS, and a neighbour whose bucket is one of them is looked up there by bisection:
// found = its id, not found = it does not exist (a key has one bucket).  Only the neighbours of other buckets (one in eight) go to
// the dictionary in HBM; the eight lanes of a k1-mer make those look-ups together as before, a wavefront skips the look-up
// numbers none of its eight k1-mers needs.  Blocks whose buckets hold more than REC_CAP keys take the dictionary for everything.
#define REC_G 4
#define REC_CAP 2048
#ifndef REC_FLY
#define REC_FLY 2
#endif
__global__ __launch_bounds__(256) void ext_records_lds_kernel(const TabIdx T, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ weight,
                                                              uint64_t n_buckets, int k, int canonical, Rec* __restrict__ rec,
                                                              const unsigned long long* __restrict__ lines, int ablate) {
  // ablate (timing experiments only, wrong records): 1 = no dictionary look-ups, 2 = no LDS bisection either, 3 = no neighbour minimizers
  __shared__ uint64_t skeys[REC_CAP];
  __shared__ uint8_t sflags[REC_CAP];
  __shared__ uint64_t sboff[REC_G + 1];
  const uint64_t* __restrict__ tkeys = T.keys;
  const uint64_t B0 = (uint64_t)blockIdx.x * REC_G;
  const uint32_t G = (uint32_t)(n_buckets - B0 < REC_G ? n_buckets - B0 : REC_G);
  if (threadIdx.x <= G) sboff[threadIdx.x] = T.boff[B0 + threadIdx.x];
  __syncthreads();
  const uint64_t lo = sboff[0];
  const uint64_t nk64 = sboff[G] - lo;
  if (nk64 == 0) return;
  const bool staged = nk64 <= REC_CAP;
  const uint32_t nk = (uint32_t)nk64;                                   // (a bucket range of 2^32 keys does not occur: the table holds fewer)
  if (staged) for (uint32_t t = threadIdx.x; t < nk; t += blockDim.x) { skeys[t] = tkeys[lo + t]; sflags[t] = flags[lo + t]; }
  __syncthreads();
  const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const int lane = threadIdx.x & 63, g0 = lane & ~7, p = lane & 7;
  const int m = T.m, w = k - m + 1;
  const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
  for (uint64_t base = 0; base < nk64; base += blockDim.x >> 3) {
    const uint64_t t = base + (threadIdx.x >> 3);
    const bool in = t < nk64;
    const uint64_t i = lo + (in ? t : 0);
    const uint64_t str = in ? (staged ? skeys[t] : tkeys[i]) : 0ULL;
    const uint8_t f = in ? (staged ? sflags[t] : flags[i]) : (uint8_t)2;
    const uint32_t wt = (in && (p == 2 || p == 6)) ? weight[i] : 0u;
    const bool dead0 = (f & 2) != 0;
    const bool dead1 = dead0 || (f & 1) || !canonical;
    uint64_t mykey; uint32_t mystrand = 0;
    {
      const uint64_t b = (uint64_t)(p & 3);
      mykey = p < 4 ? (((str << 2) | b) & mask) : ((str >> 2) | (b << (2 * (k - 1))));
      if (canonical) { const uint64_t rc = shn_revcomp(mykey, k); if (rc < mykey) { mykey = rc; mystrand = 1; } }
      if (dead0) mykey = 0;
    }
    // the neighbour's bucket from the k1-mer's own m-mers (see ext_records_kernel)
    uint32_t nbkt = (uint32_t)B0;
    if (ablate < 3) {
      uint32_t smin = 0xFFFFFFFFu, pmin = 0xFFFFFFFFu;
      for (int pos = p; pos < w; pos += 8) {
        const uint32_t fm = (uint32_t)(str >> (2 * (k - m - pos))) & mmask;
        uint32_t c = fm;
        if (canonical) { const uint32_t r = shn_revcomp32(fm, m); c = r < fm ? r : fm; }
        const uint32_t o = shn_sk_order(c);
        if (pos >= 1) smin = o < smin ? o : smin;
        if (pos <= w - 2) pmin = o < pmin ? o : pmin;
      }
#pragma unroll
      for (int d = 1; d < 8; d <<= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)smin, d, 64), b2 = (uint32_t)__shfl_xor((int)pmin, d, 64);
        smin = a < smin ? a : smin; pmin = b2 < pmin ? b2 : pmin;
      }
      const uint32_t nb = (uint32_t)(p & 3);
      const uint32_t fm = p < 4 ? ((((uint32_t)str & (mmask >> 2)) << 2) | nb)
                                : ((nb << (2 * (m - 1))) | ((uint32_t)(str >> (2 * (k - m + 1))) & (mmask >> 2)));
      uint32_t c = fm;
      if (canonical) { const uint32_t r = shn_revcomp32(fm, m); c = r < fm ? r : fm; }
      uint32_t o = shn_sk_order(c);
      const uint32_t shared = p < 4 ? smin : pmin;
      o = shared < o ? shared : o;
      nbkt = shn_sk_bucket(o, T.bits);
    }
    bool cross = !dead0;                                                // this lane's look-up goes to the dictionary
    uint32_t w_own = 0xFFFFFFFFu;
    if (ablate >= 2) { cross = false; w_own = (uint32_t)mykey & 0x7FFFFFFu; }
    else if (!dead0 && staged && (uint64_t)nbkt >= B0 && (uint64_t)nbkt < B0 + G) {
      cross = false;
      uint32_t a = (uint32_t)(sboff[nbkt - B0] - lo), b = (uint32_t)(sboff[nbkt - B0 + 1] - lo);
      while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        const uint64_t v = skeys[mid];
        if (v == mykey) { const uint8_t fj = sflags[mid]; if (!(fj & 2)) w_own = (uint32_t)(lo + mid) | ((fj & 1) ? FD_PAL : 0u); break; }
        if (v < mykey) a = mid + 1; else b = mid;
      }
    }
    if (ablate == 1) cross = false;
    const uint64_t myline = cross ? fd_line_in_bucket(T, nbkt, mykey) : 0ULL;
    const uint32_t strands = (uint32_t)((__ballot(mystrand != 0) >> g0) & 0xFFULL);
    uint32_t need = 0;                                                  // look-up numbers some k1-mer of this wavefront sends to the dictionary
#pragma unroll
    for (int q = 0; q < 8; q++) if (__ballot(cross && p == q)) need |= 1u << q;
    const uint64_t ckey = cross ? mykey : 0ULL;
    uint32_t r8[8], d8[8];
#pragma unroll
    for (int h = 0; h < 8; h += REC_FLY) {                              // (REC_FLY dictionary look-ups in flight at a time: one in eight is a real one)
      ulonglong2 v[REC_FLY];
#pragma unroll
      for (int q = 0; q < REC_FLY; q++)
        if ((need >> (h + q)) & 1) v[q] = ((const ulonglong2*)(lines + shfl_u64(myline, g0 + h + q) * 16))[p];
#pragma unroll
      for (int q = 0; q < REC_FLY; q++) {
        uint32_t wq = (uint32_t)__shfl((int)w_own, g0 + h + q, 64);
        if ((need >> (h + q)) & 1) {
          const uint64_t kq = shfl_u64(ckey, g0 + h + q);
          const uint32_t wd = fd_match(v[q], lines, shfl_u64(myline, g0 + h + q), kq, p, g0, T, flags);
          if (kq != 0) wq = wd;                                         // (this group's look-up was one for the dictionary)
        }
        if (wq == 0xFFFFFFFFu) { r8[h + q] = d8[h + q] = 0xFFFFFFFFu; continue; }
        const uint32_t j = wq & ~FD_PAL;
        const uint32_t st = (strands >> (h + q)) & 1u;
        r8[h + q] = 2 * j + st;
        d8[h + q] = (wq & FD_PAL) ? r8[h + q] : 2 * j + (1 - st);
      }
    }
    if (in) {
      Quad out;
      const Quad none = Quad{~0u, ~0u, ~0u, ~0u};
      switch (p) {
        case 0: out = Quad{r8[0], r8[1], r8[2], r8[3]}; break;
        case 1: out = Quad{r8[4], r8[5], r8[6], r8[7]}; break;
        case 4: out = dead1 ? none : Quad{d8[7], d8[6], d8[5], d8[4]}; break;
        case 5: out = dead1 ? none : Quad{d8[3], d8[2], d8[1], d8[0]}; break;
        case 2: case 6: out = Quad{wt, NOHINT_WORD, 0xFFFFFFFFu, 0u}; break;
        default: out = Quad{0u, 0u, 0u, 0u}; break;
      }
      ((Quad*)(rec + 2 * i))[p] = out;
    }
  }
}

// ---- connected components of the k1-mer graph (vertices = canonical k1-mers, edges = the adjacency rows).
// A walk never leaves its component, so the components can be extended independently -- on different GPUs.
// Lock-free union-find: roots only ever link to smaller ids (no cycles), finds halve paths as they go.
__device__ __forceinline__ uint32_t cc_find(uint32_t* lab, uint32_t x) {
  uint32_t cur = x;
  while (true) {
    uint32_t p = __hip_atomic_load(&lab[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == cur) return cur;
    uint32_t gp = __hip_atomic_load(&lab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) __hip_atomic_store(&lab[cur], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cur = p;
  }
}
__global__ void cc_init_kernel(uint32_t* __restrict__ lab, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lab[i] = (uint32_t)i;
}
__device__ __forceinline__ void cc_unite(uint32_t* lab, uint32_t u, uint32_t v) {
  while (true) {
    const uint32_t ru = cc_find(lab, u), rv = cc_find(lab, v);
    if (ru == rv) return;
    const uint32_t hi = ru > rv ? ru : rv, lo = ru > rv ? rv : ru;
    if (atomicCAS(&lab[hi], hi, lo) == hi) return;
  }
}
// Every edge of the k1-mer graph straight from the table, no adjacency rows in between: one thread per (canonical k1-mer, which),
// which = 0..7: the eight neighbours of its forward orientation (append / prepend a base: the other orientation has the same
// ones), which = 8..15: its siblings.  contig_connections joins contigs that share a K-mer (extension_correction.py:372-390):
// besides adjacent k1-mers those are k1-mers with the same K-suffix (x.m, x'.m) or the same K-prefix (m.y, m.y') -- not adjacent,
// and only joined through a common neighbour if that neighbour exists and is not low-complexity (a transcript's last K-mer before
// a poly-A tail is the typical exception).  So the labelling also unites every k1-mer with its (up to six) siblings.
// Look-ups as in the records kernel: eight lanes per canonical k1-mer, through the one-line dictionary.
// diagnostics (round 5): look-ups of the labelling kernel that found their key / unions made / unions that found both ends united already
__device__ unsigned long long g_cc_dbg[4];
extern "C" int shn_debug_cc_counters(uint64_t* out4, int reset) {
  unsigned long long h[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cc_dbg), sizeof(h)) != hipSuccess) return shn_fail(SHN_ERR_HIP, "shn_debug_cc_counters");
  for (int i = 0; i < 4; i++) out4[i] = h[i];
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_cc_dbg), z, sizeof(z)) != hipSuccess) return shn_fail(SHN_ERR_HIP, "shn_debug_cc_counters"); }
  return SHN_OK;
}
static int cc_half_mode() { return ((getenv("SHN_CC_HALF") && getenv("SHN_CC_HALF")[0] == '0') ? 0 : 1) | (getenv("SHN_CC_DEBUG") ? 2 : 0); }
__global__ void cc_edges_kernel(const TabIdx T, const uint8_t* __restrict__ flags,
                                uint64_t n, int k, int canonical, uint32_t* lab, const unsigned long long* __restrict__ lines, uint64_t n_lines, int half_mode) {
  const uint64_t* __restrict__ tkeys = T.keys;
  const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const uint64_t total = n * 8, rounded = (total + 63) & ~63ULL;
  const int lane = threadIdx.x & 63, g0 = lane & ~7, p = lane & 7;
  for (uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < rounded; gid += (uint64_t)gridDim.x * blockDim.x) {
    const bool in = gid < total;
    const uint64_t i = in ? gid >> 3 : 0;
    const bool dead = !in || (flags[i] & 2);
    const uint64_t str = tkeys[i];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      uint64_t mykey;
      {
        const uint64_t b = (uint64_t)(p & 3);
        bool skip = dead;
        if (half == 0) mykey = (p & 4) ? ((str >> 2) | (b << (2 * (k - 1)))) : (((str << 2) | b) & mask);
        else if (p & 4) { skip |= (str & 3) == b; mykey = (str & ~3ULL) | b; }
        else { const int sh = 2 * (k - 1); skip |= ((str >> sh) & 3) == b; mykey = (str & ~(3ULL << sh)) | (b << sh); }
        if (canonical) { const uint64_t rc = shn_revcomp(mykey, k); if (rc < mykey) mykey = rc; }
        // (every edge is seen from both of its ends -- the neighbour and sibling relations are symmetric, and so is "both not
        // low-complexity" -- so an end asks only for the larger keys: half the look-ups; what is not asked goes to line 0, which the
        // caches hold)
        if (skip || ((half_mode & 1) && mykey <= str)) mykey = 0;
      }
      const uint64_t myline = mykey ? fd_bucket(T, mykey, n_lines) : 0ULL;
      uint64_t key[8];
      ulonglong2 v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        key[q] = shfl_u64(mykey, g0 + q);
        v[q] = ((const ulonglong2*)(lines + shfl_u64(myline, g0 + q) * 16))[p];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t w = fd_match(v[q], lines, shfl_u64(myline, g0 + q), key[q], p, g0, T, flags);
        // (lane q of the group does the union: eight independent ones side by side)
        if (w != 0xFFFFFFFFu && p == q && (uint64_t)(w & ~FD_PAL) != i) {
          if (half_mode & 2) atomicAdd(&g_cc_dbg[0], 1ULL);
          cc_unite(lab, (uint32_t)i, w & ~FD_PAL);
        }
      }
    }
  }
}
// Every k1-mer gets its root.  The find here must NOT compress: a compressing find of one thread stores an ancestor into lab[j]
// (correct inside the union-find, where any ancestor will do) -- and when that store lands after thread j has written j's root, j
// keeps a label that is not a root.  Found in round 5 when the labelling asked every edge from one end only: the trees were deeper
// at this point, 70 % of the runs left 1-40 k1-mers of a 227 k table with an ancestor for a label (a k1-mer then went to another
// rank than its component).  With every edge united twice the trees are all but flat here and the window almost never opened --
// almost.  Without stores other than the roots themselves, every value a find can read is an ancestor and the roots do not move.
__device__ __forceinline__ uint32_t cc_find_readonly(const uint32_t* lab, uint32_t x) {
  uint32_t cur = x;
  while (true) {
    const uint32_t p = __hip_atomic_load(&lab[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == cur) return cur;
    cur = p;
  }
}
__global__ void cc_flatten_kernel(uint32_t* lab, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t r = cc_find_readonly(lab, (uint32_t)i); __hip_atomic_store(&lab[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
// size estimate of the components from every 64th k1-mer (a full count would hammer a handful of addresses)
__global__ void cc_sample_kernel(const uint32_t* __restrict__ lab, uint64_t n, uint32_t* __restrict__ size_s) {
  uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 64;
  if (i < n) atomicAdd(&size_s[lab[i]], 1u);
}
// owner of every root by hash; roots of big components are listed for the host to balance
__global__ void cc_owner_kernel(const uint32_t* __restrict__ lab, const uint32_t* __restrict__ size_s, uint64_t n, uint32_t world,
                                uint8_t* __restrict__ owner_root, uint32_t* __restrict__ big_root, uint32_t* __restrict__ big_size,
                                unsigned long long* __restrict__ n_big, uint32_t big_cap) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (lab[i] != (uint32_t)i) { owner_root[i] = 0xFF; return; }
  owner_root[i] = (uint8_t)(shn_mix64((uint64_t)i ^ 0x5851F42D4C957F2DULL) % world);
  if (size_s[i] >= 16) {
    unsigned long long p = atomicAdd(n_big, 1ULL);
    if (p < big_cap) { big_root[p] = (uint32_t)i; big_size[p] = size_s[i]; }
  }
}
__global__ void cc_assign_kernel(const uint32_t* __restrict__ big_root, const uint8_t* __restrict__ big_owner, uint32_t n_big,
                                 uint8_t* __restrict__ owner_root) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_big) owner_root[big_root[j]] = big_owner[j];
}

// The seeds (oriented k1-mers of weight >= min_weight; low-complexity ones and the second orientation of a palindrome left out) in
// two passes without a shared cursor: the count pass leaves one count per block, a scan turns them into bases, the write pass
// recomputes and writes -- in table order, whatever the scheduling.  (One atomic per block on ONE address was 17 ms per pass at
// 1.4 M blocks.)  One thread per canonical k1-mer, both orientations.
__global__ __launch_bounds__(1024) void ext_seed_kernel(const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ weight,
                                const uint8_t* __restrict__ flags, uint64_t n, int k, int canonical, uint32_t min_weight,
                                uint32_t* __restrict__ block_count, const uint64_t* __restrict__ block_base,
                                uint64_t* __restrict__ skeys, uint32_t* __restrict__ svals) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool s0 = false, s1 = false;
  if (i < n) {
    const uint8_t f = flags[i];
    s0 = !(f & 2) && weight[i] >= min_weight;
    s1 = s0 && !(f & 1) && canonical;
  }
  __shared__ uint32_t wcnt[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1), below = (1ULL << lane) - 1ULL;
  if (lane == 0) wcnt[wid] = (uint32_t)(__popcll(m0) + __popcll(m1));
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) { const uint32_t c = wcnt[w]; if (w < wid) before += c; total += c; }
  if (!skeys) { if (threadIdx.x == 0) block_count[blockIdx.x] = total; return; }
  if (!s0) return;
  uint64_t p = block_base[blockIdx.x] + before + (uint32_t)(__popcll(m0 & below) + __popcll(m1 & below));
  const uint64_t key = tkeys[i];
  skeys[p] = key; svals[p] = (uint32_t)(2 * i);
  if (s1) { skeys[p + 1] = shn_revcomp(key, k); svals[p + 1] = (uint32_t)(2 * i + 1); }
}

__global__ void ext_weightkey_kernel(const uint32_t* __restrict__ svals, const uint32_t* __restrict__ weight, uint64_t ns,
                                     uint64_t* __restrict__ wkeys) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  wkeys[i] = (uint64_t)(0xFFFFFFFFu - weight[svals[i] >> 1]);   // ascending sort => weight descending
}

// entry b of a row, b known only at run time: picked with compares (an indexed access makes the compiler keep the row in
// scratch memory)
__device__ __forceinline__ int32_t adj_get(const Adj4& a, int b) { return b == 0 ? a.v[0] : b == 1 ? a.v[1] : b == 2 ? a.v[2] : a.v[3]; }
#define CHUNK_SHIFT 4          // 16 oriented k1-mers = one 128-byte line of claims per flag

struct WalkArgs {
  const uint32_t* order; RowView adjR; RowView adjL; WordView weight;   // weight: by ORIENTED id (views of the record array)
  u64* claim;            // live claims: clean walks' + this round's (dirty walks released theirs before the round)
  const u64* claim_old;  // snapshot taken before the round
  uint32_t* nr_out; uint32_t* nl_out; uint64_t* totw_out;
  // memo: the path of the walk's last live run, rebuilt from the claims after every round it ran alive
  // (ext_memo_plan_kernel + the scatter in ext_mark_kernel).  Hints only -- every use is validated.
  const uint32_t* pool; const uint64_t* moff; const uint32_t* mR; const uint32_t* mL; const uint8_t* mvalid;
  WordView hint;         // per k1-mer: where it was last written into a memo (pool index << 2 | kind), NOHINT if never
  unsigned long long* steps_counter;
  unsigned long long* fresh_steps_counter;   // ... of them in the first round of a rank block (ext_walk_kernel<true>: its own line in bench.py's kernel table)
  unsigned long long* wave_steps_counter;
  unsigned long long* dbg;     // [0] wave steps confirmed from an own memo [1] from a foreign memo
  // a thread walker that turns out long hands its walk over to a wavefront (same round): where it stands
  uint32_t* promo_list; unsigned long long* promo_count; uint32_t* res_cur; uint32_t* res_info;   // info = dir << 31 | steps so far
  uint32_t promote_steps;
  uint8_t* chunk;        // per 2^CHUNK_SHIFT oriented k1-mers: "a claim in here was written this round" (the mark pass visits only those)
  uint8_t* robbed;       // per walk: a claim of its record is not (or no longer) its own -- see note_claim
  int seed_check;        // thread walkers: look at the own seed's claim on every step and stop when a lower rank has taken it
  int first_look;        // thread walkers: a direction starts with a look at the candidates' claims alone
  // claim logs (round 6): in a bulk round -- no memos are made there -- the thread walker writes the k1-mers it claims into a chain of
  // 8-word chunks of its own ([0] = the chunk before, [1..7] = k1-mers; the seed is not logged: it is order[r]); a walk that re-runs
  // later gives back what it still holds through its log (ext_release_memo_kernel) instead of the begin pass streaming every claim
  uint32_t* logpool; uint32_t* log_head; uint8_t* log_cnt; unsigned long long* log_cursor; uint64_t log_cap;   // pool of log_cap chunks; per walk: last chunk (NONE32: no log, LOG_LOST: the pool ran out) and its entries
};
#define LOG_WORDS 8u
#define LOG_PER (LOG_WORDS - 1u)
#define LOG_SLAB 64u           // chunks a wavefront reserves with one global atomic
#define LOG_LOST 0xFEFEFEFEu        // (the byte pattern of hipMemset(0xFE): "no log" is what the arrays start as)

// Claim `node` as step `pos` of walk r: atomic min on rank:pos.  Returns what stood there before; the caller hands it to note_claim
// -- one step later in the thread walkers, where the answer has long arrived behind the loads of the next step (memory operations
// of a wavefront return in issue order), so the atomic's round trip is on nobody's critical path.
__device__ __forceinline__ u64 claim_node(const WalkArgs& A, uint32_t node, uint32_t r, uint32_t pos) {
  const u64 old = __hip_atomic_fetch_min(&A.claim[node], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (A.chunk) A.chunk[node >> CHUNK_SHIFT] = 1;
  return old;
}
// What a claim found: a lower rank (it got there between this walk's look and its claim: the walk's record holds a k1-mer it does
// not own) or a higher one (robbed: THAT walk's record is stale, if it ran this round -- one that sat out is found by the mark
// pass through the snapshot).  Either way the walk has to run again: ext_verify_kernel reads the flags.  (Until round 3 the mark
// pass counted the k1-mers every walk owned, one random atomic per claimed k1-mer, and the verify kernel compared the count with
// the record.)
__device__ __forceinline__ void note_claim(const WalkArgs& A, u64 old, uint32_t r) {
  const uint32_t x = RANK(old);
  if (x != UNCLAIMED && x != r) A.robbed[x < r ? r : x] = 1;
}

// One greedy decision (extension_correction.py:223-237): among the candidates that exist and are not
// traversed pick the heaviest, ties in BASES order A,G,C,T (codes 0,2,1,3; strict >).  Traversed = claimed
// live by a rank <= r (lower ranks of this round, or this walk's own trail), or claimed in the pre-round
// snapshot by a lower rank.
__device__ __forceinline__ int decide(const Adj4& cand, uint32_t r, const u64* claim, const u64* __restrict__ claim_old,
                                      const WordView weight, uint32_t dummy, uint32_t& bw) {
  u64 cl[4], co[4];
  uint32_t w[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    uint32_t idx = cand.v[b] < 0 ? dummy : (uint32_t)cand.v[b];
    cl[b] = __hip_atomic_load(&claim[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    co[b] = claim_old[idx];
    w[b] = weight[idx];
  }
  int best = -1;
  bw = 0;
#define CONSIDER(b) if (cand.v[b] >= 0 && RANK(cl[b]) > r && RANK(co[b]) >= r && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }
  CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
  return best;
}

// the greedy choice of walk r standing at step p, from the claims alone (for walk r a k1-mer is traversed iff a lower rank
// owns it or r owns it at a step <= p); used by the precise marks of ext_mark_kernel and by the fixpoint audit
__device__ __forceinline__ int audit_decide(const Adj4& cd, uint32_t r, uint32_t p, const u64* __restrict__ claim,
                                            const WordView weight) {
  int best = -1;
  uint32_t bw = 0;
#pragma unroll
  for (int bi = 0; bi < 4; bi++) {
    const int b = bi == 0 ? 0 : bi == 1 ? 2 : bi == 2 ? 1 : 3;          // BASES order A,G,C,T
    if (cd.v[b] < 0) continue;
    const u64 c = claim[cd.v[b]];
    const bool avail = RANK(c) > r || (RANK(c) == r && POS(c) > p);
    const uint32_t w = weight[(uint32_t)cd.v[b]];
    if (avail && (best < 0 || w > bw)) { best = b; bw = w; }
  }
  return best;
}


// ---- short walks: one thread per walk, one memory round trip per step (candidate rows prefetched).
// FRESH: the first round of a block that has just opened -- none of its walks has run before, so the snapshot holds nothing but
// the claims of final walks, which the live claims hold too: the snapshot is not read (a quarter of a step's memory accesses,
// in the rounds that make most of the steps).
template <bool FRESH>
__global__ __launch_bounds__(WBLK) void ext_walk_kernel(WalkArgs A, uint64_t n_walks, const uint32_t* __restrict__ list,
                                                        const u64* __restrict__ snap) {
  __shared__ unsigned long long blk_steps;
  __shared__ uint32_t blk_promo[WBLK], n_promo, promo_base;      // walks handed over: one global atomic per block
  __shared__ uint32_t slab_next, slab_end;                        // claim logs: the wavefront's reserve of chunks
  if (threadIdx.x == 0) { blk_steps = 0; n_promo = 0; slab_next = 0; slab_end = 0; }
  __syncthreads();
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mysteps = 0;
  // a chunk for every lane that asks in this trip (the lanes of the wavefront that are in the loop right now): one LDS update by
  // the first of them, one global atomic per LOG_SLAB chunks
  auto log_chunk = [&](bool need) -> uint32_t {
    const unsigned long long m = __ballot(need);
    uint32_t got = NONE32;
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      const uint32_t cnt = (uint32_t)__popcll(m);
      uint32_t base = 0;
      if ((int)threadIdx.x == leader) {
        uint32_t nx = *(volatile uint32_t*)&slab_next, en = *(volatile uint32_t*)&slab_end;
        if (nx + cnt > en) {
          const unsigned long long g = atomicAdd(A.log_cursor, (unsigned long long)LOG_SLAB);
          if (g + LOG_SLAB <= A.log_cap) { nx = (uint32_t)g; en = nx + LOG_SLAB; } else { nx = NONE32 - 2 * LOG_SLAB; en = nx; }     // (the pool ran out: nobody gets a chunk)
        }
        base = nx + cnt <= en ? nx : NONE32;
        *(volatile uint32_t*)&slab_next = nx + (base == NONE32 ? 0u : cnt); *(volatile uint32_t*)&slab_end = en;
      }
      base = (uint32_t)__shfl((int)base, leader, 64);
      if (need && base != NONE32) got = base + (uint32_t)__popcll(m & ((1ULL << threadIdx.x) - 1ULL));
    }
    return got;
  };
  const bool logging = A.logpool != nullptr;
  uint32_t lg_chunk = NONE32, lg_k = LOG_PER;                     // the walk's last chunk and the entries it holds (LOG_PER: full, or none yet)
  bool lg_lost = false;
  if (t < n_walks) {
    const uint32_t r = list[t];
    const uint32_t o = A.order[r];
    const unsigned long long t_begin = A.dbg ? __builtin_amdgcn_s_memrealtime() : 0ULL;      // (100 MHz)
    uint32_t nr = 0, nl = 0;
    uint64_t tot = 0;
    bool promoted = false;
    // snap: the pre-round snapshot; A.claim: live claims of this round
    bool isvoid = (!FRESH && RANK(snap[o]) < r) || RANK(__hip_atomic_load(&A.claim[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r;
    if (!isvoid) {
      u64 seen = claim_node(A, o, r, 0);             // what the last claim found; looked at one step later
      uint32_t pos = 0, pend = NONE32;
      tot = A.weight[o];
      bool gave_up = false;
      for (int dir = 0; dir < 2; dir++) {
        const RowView adj = dir == 0 ? A.adjR : A.adjL;
        uint32_t steps = 0;
        Adj4 cand = adj[o];
        // SHN_EXT_FIRST_LOOK=1 (experiment, off): a walk that cannot take a single step finds that out from its candidates' CLAIMS alone
        // -- up to four lines -- instead of their claims, weights and rows (eight).  No gain at BASELINE configs[2] (HISTORY.md, Round 5).
        if (A.first_look) {
          bool any = false;
#pragma unroll
          for (int b = 0; b < 4; b++)
            if (cand.v[b] >= 0) {
              const u64 c0 = __hip_atomic_load(&A.claim[cand.v[b]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const u64 f0 = FRESH ? UNCLAIMED64 : snap[cand.v[b]];
              any |= RANK(c0) > r && RANK(f0) >= r;
            }
          if (!any) continue;                      // (no step in this direction: nr / nl stay 0)
        }
        while (true) {
          u64 cl[4], cf[4];
          uint32_t w[4];
          Adj4 nxt[4];
#pragma unroll
          for (int b = 0; b < 4; b++) {
            uint32_t idx = cand.v[b] < 0 ? o : (uint32_t)cand.v[b];
            cl[b] = __hip_atomic_load(&A.claim[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cf[b] = FRESH ? UNCLAIMED64 : snap[idx];
            w[b] = A.weight[idx];
            nxt[b] = adj[idx];
          }
          // the walk's own seed (one line, in the L2 from the second step on): once a lower rank has taken it this walk is void in
          // the end -- 98.6 % of the walks of BASELINE configs[2] are -- and whatever it goes on to claim is wasted; it stops, is
          // flagged like any robbed walk and looks again next round
          // (seed_check 2 -- experiment: the walk's robbed flag instead, set by whoever took ANY of its k1-mers)
          const u64 cseed = A.seed_check == 1 ? __hip_atomic_load(&A.claim[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : A.seed_check == 2 ? (*(volatile const uint8_t*)&A.robbed[r] ? 0ULL : CLAIM(r, 0)) : CLAIM(r, 0);
          // the claim of the step just taken goes out BEHIND the loads of this step: vector memory operations of a wavefront
          // return in issue order (one counter for loads, stores and atomics), so a claim issued in front of the loads would put
          // the latency of a memory-side atomic on every step of the walk -- and a bulk round lasts as long as its longest walk
          u64 found = UNCLAIMED64;
          if (pend != NONE32) {
            found = claim_node(A, pend, r, pos);
            pend = NONE32;
          }
          int best = -1;
          uint32_t bw = 0;
#define CONSIDER(b) if (cand.v[b] >= 0 && RANK(cl[b]) > r && RANK(cf[b]) >= r && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }
          CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
          // (what the claim BEFORE this step's found: it was issued in front of this step's loads, which have just been used)
          note_claim(A, seen, r);
          seen = found;
          if (RANK(cseed) < r) { A.robbed[r] = 1; gave_up = true; break; }
          if (best < 0) break;
          // (selected with compares, not indexed: a run-time index puts the rows into scratch memory -- 96 bytes per lane of private
          // memory traffic on every step)
          uint32_t nbest = (uint32_t)(best == 0 ? cand.v[0] : best == 1 ? cand.v[1] : best == 2 ? cand.v[2] : cand.v[3]);
          pos++;
          pend = nbest;                            // claimed at the top of the next trip (or below, when the walk stops here)
          steps++;
          tot += bw;
          if (logging) {                           // the k1-mer goes into the walk's log (whoever gets this far claims it)
            const bool need = lg_k == LOG_PER && !lg_lost;
            const uint32_t nc = log_chunk(need);
            if (need) {
              if (nc == NONE32) lg_lost = true;
              else { A.logpool[(uint64_t)nc * LOG_WORDS] = lg_chunk; lg_chunk = nc; lg_k = 0; }
            }
            if (!lg_lost) { A.logpool[(uint64_t)lg_chunk * LOG_WORDS + 1 + lg_k] = nbest; lg_k++; }
          }
          if (pos >= A.promote_steps) {            // long after all: a wavefront takes over from here (memos, 64 steps a trip)
            note_claim(A, seen, r);
            seen = claim_node(A, nbest, r, pos);
            pend = NONE32;
            A.res_cur[r] = nbest;
            A.res_info[r] = ((uint32_t)dir << 31) | pos;
            blk_promo[atomicAdd(&n_promo, 1u)] = r;
            promoted = true;
            break;
          }
#pragma unroll
          for (int q = 0; q < 4; q++) cand.v[q] = best == 0 ? nxt[0].v[q] : best == 1 ? nxt[1].v[q] : best == 2 ? nxt[2].v[q] : nxt[3].v[q];   // (word by word: a select between structs goes through memory)
        }
        if (pend != NONE32) {                        // the last step of this direction
          note_claim(A, seen, r);
          seen = claim_node(A, pend, r, pos);
          pend = NONE32;
        }
        if (dir == 0) nr = steps; else nl = steps;
        if (promoted || gave_up) break;
      }
      note_claim(A, seen, r);
    }
    A.nr_out[r] = isvoid ? UNCLAIMED : nr;
    A.nl_out[r] = nl;
    A.totw_out[r] = tot;
    if (logging) { A.log_head[r] = lg_lost ? LOG_LOST : lg_chunk; A.log_cnt[r] = (uint8_t)(lg_chunk == NONE32 ? 0u : lg_k); }      // (a walk that took no step: NONE32 -- its seed is all it holds)
    mysteps = nr + nl;
    // debug: the longest walk of the launch and how long it took (steps << 32 | ticks of 10 ns): a bulk round cannot end before it
    if (A.dbg && mysteps >= 64) atomicMax(&A.dbg[12], ((unsigned long long)mysteps << 32) | ((__builtin_amdgcn_s_memrealtime() - t_begin) & 0xFFFFFFFFULL));
  }
  if (mysteps) atomicAdd(&blk_steps, (unsigned long long)mysteps);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (blk_steps) { atomicAdd(A.steps_counter, blk_steps); if (FRESH) atomicAdd(A.fresh_steps_counter, blk_steps); }
    promo_base = n_promo ? (uint32_t)atomicAdd(A.promo_count, (unsigned long long)n_promo) : 0u;
  }
  __syncthreads();
  if (threadIdx.x < n_promo) A.promo_list[promo_base + threadIdx.x] = blk_promo[threadIdx.x];
}

// ---- bulk rounds, second half: the walks the thread kernel handed over after `promote_steps` steps (the long ones: a few per cent
// of a bulk round's walks, nearly all of its steps), packed.  In the launch above a wavefront lasts as long as its longest walk while
// 63 of its 64 walks end within a few steps (98.6 % of the walks of BASELINE configs[2] are void in the end): one lane busy in
// sixty.  Here every lane is a long walk, and a lane whose walk ends takes the next one from the list (one flat loop: an iteration
// is one step of every lane's walk; the lanes that need a walk fetch one at the top of the iteration, one atomic for all of them),
// so a wavefront is never held by one 2,000-step walk.  The step itself is ext_walk_kernel's, statement for statement.
template <bool FRESH>
__global__ __launch_bounds__(WBLK) void ext_walk_resume_kernel(WalkArgs A, const uint32_t* __restrict__ list, const unsigned long long* __restrict__ list_count,
                                                               unsigned long long* __restrict__ head, const u64* __restrict__ snap) {
  const unsigned long long n_list = *list_count;
  const int lane = threadIdx.x & 63;
  bool have = false, out_of_work = false;
  uint32_t r = 0, o = 0, pos = 0, steps = 0, nr = 0, pend = NONE32, walked = 0;
  int dir = 0;
  uint64_t tot = 0;
  u64 seen = UNCLAIMED64;
  Adj4 cand = {{-1, -1, -1, -1}};
  unsigned long long my_steps = 0;
  while (true) {
    // lanes without a walk take the next ones of the list
    const unsigned long long want = __ballot(!have && !out_of_work);
    if (want) {
      const int leader = __ffsll((long long)want) - 1;
      unsigned long long base = 0;
      if (lane == leader) base = atomicAdd(head, (unsigned long long)__popcll(want));
      base = shfl_u64(base, leader);
      if (!have && !out_of_work) {
        const unsigned long long idx = base + (unsigned long long)__popcll(want & ((1ULL << lane) - 1ULL));
        if (idx >= n_list) out_of_work = true;
        else {
          r = list[idx];
          o = A.order[r];
          const uint32_t info = A.res_info[r];
          dir = (int)(info >> 31);
          pos = info & 0x7FFFFFFFu;
          const uint32_t cur = A.res_cur[r];
          nr = dir ? A.nr_out[r] : 0u;
          steps = dir ? pos - nr : pos;
          tot = A.totw_out[r];
          cand = (dir == 0 ? A.adjR : A.adjL)[cur];
          pend = NONE32; seen = UNCLAIMED64; walked = 0;
          have = true;
        }
      }
    }
    if (!__ballot(have)) break;
    if (!have) continue;
    // ---- one step (ext_walk_kernel's loop body)
    const RowView adj = dir == 0 ? A.adjR : A.adjL;
    u64 cl[4], cf[4];
    uint32_t w[4];
    Adj4 nxt[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t idx = cand.v[b] < 0 ? o : (uint32_t)cand.v[b];
      cl[b] = __hip_atomic_load(&A.claim[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cf[b] = FRESH ? UNCLAIMED64 : snap[idx];
      w[b] = A.weight[idx];
      nxt[b] = adj[idx];
    }
    const u64 cseed = A.seed_check == 1 ? __hip_atomic_load(&A.claim[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : A.seed_check == 2 ? (*(volatile const uint8_t*)&A.robbed[r] ? 0ULL : CLAIM(r, 0)) : CLAIM(r, 0);
    u64 found = UNCLAIMED64;
    if (pend != NONE32) { found = claim_node(A, pend, r, pos); pend = NONE32; }
    int best = -1;
    uint32_t bw = 0;
#define CONSIDER(b) if (cand.v[b] >= 0 && RANK(cl[b]) > r && RANK(cf[b]) >= r && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }
    CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
    note_claim(A, seen, r);
    seen = found;
    bool done = false;
    if (RANK(cseed) < r) { A.robbed[r] = 1; done = true; }                       // the seed went to a lower rank: void in the end, stop here
    else if (best < 0) {
      if (dir == 0) { nr = steps; dir = 1; steps = 0; cand = A.adjL[o]; }       // the right end: on to the left, from the seed
      else done = true;
    } else {
      const uint32_t nbest = (uint32_t)(best == 0 ? cand.v[0] : best == 1 ? cand.v[1] : best == 2 ? cand.v[2] : cand.v[3]);
      pos++; pend = nbest; steps++; walked++; tot += bw;
#pragma unroll
      for (int q = 0; q < 4; q++) cand.v[q] = best == 0 ? nxt[0].v[q] : best == 1 ? nxt[1].v[q] : best == 2 ? nxt[2].v[q] : nxt[3].v[q];
    }
    if (done) {
      note_claim(A, seen, r);
      const uint32_t nl = dir == 0 ? 0u : steps;
      if (dir == 0) nr = steps;
      A.nr_out[r] = nr; A.nl_out[r] = nl; A.totw_out[r] = tot;
      my_steps += walked;
      have = false;
    }
  }
  for (int off = 32; off > 0; off >>= 1) my_steps += shfl_u64(my_steps, lane ^ off);
  if (lane == 0 && my_steps) atomicAdd(A.steps_counter, my_steps);              // (thread-walker steps, like the ones before the hand-over)
}

// ---- long walks: one wavefront per dirty walk.  A memo (the path of some walk's last live run, own or foreign)
// is re-checked 64 steps per memory round trip; the walk is sequential only from the first changed decision
// until it meets a memo again -- its own, or the one of the walk whose territory it is taking over.
// Memo entries are hints: an entry counts only if the k1-mer's hint says it sits at exactly that position of
// that memo (so the validated entries of a chunk are pairwise distinct), and a changed decision is re-made
// sequentially against the live claims, never taken from the speculative lane.
__device__ __forceinline__ bool is_term_any(bool term, bool at_mark) { return term && at_mark; }
// Memo slots are never rewritten: a walk that runs again gets a new slot, so a hint always leads to an intact old path.
//   slot = [MARK][nR | HI][seed][R_1 .. R_nR][MARK][L_1 .. L_nL][MARK]       (HI = top bit; k1-mer ids are < 2^31)
// Every word with the top bit set ends a segment: marks, the header, and the NONE32 holes of steps that were robbed.
#define MEMO_MARK 0xFFFFFFFEu
#define MEMO_HI 0x80000000u
#define NOHINT 0xFFFFFFFFu
#define HINT_R 0u
#define HINT_L 1u
#define HINT_SEED 2u
struct MemoCursor { int64_t i; int32_t step; bool term; };
// i = pool index of the next expected step, step = +1 / -1 (a memo can be followed against the direction its walk
// took), term = the segment ends where that walk ended (then "stop" is the expected decision at its mark)

// `node` was just reached going in direction dir; its hint says where it sits in some memo: follow that memo
struct WhyStat { uint32_t c2 = 0, c4 = 0, c5 = 0, c7 = 0; };      // (named fields, not an array: an array whose address is passed on lives in scratch)
__device__ __forceinline__ bool memo_follow(const WalkArgs& A, uint32_t hh, uint32_t node, int dir, MemoCursor& mc, WhyStat* why = nullptr) {
#define WHY(i) do { if (A.dbg && threadIdx.x == 0) atomicAdd(&A.dbg[i], 1ULL); if (why) why->c##i++; } while (0)
  if (hh == NOHINT) { WHY(2); return false; }                     // never written into a memo
  const int64_t idx = (int64_t)(hh >> 2);
  const uint32_t kind = hh & 3u;
  if (A.pool[idx] != node) { WHY(4); return false; }              // (cannot happen while slots are not recycled)
  if (kind == HINT_SEED) {                                          // the walk's part of my direction, forwards
    const uint32_t nR = A.pool[idx - 1] & ~MEMO_HI;
    mc.i = dir == 0 ? idx + 1 : idx + 2 + (int64_t)nR; mc.step = 1; mc.term = true;
  } else if ((kind == HINT_R) == (dir == 0)) { mc.i = idx + 1; mc.step = 1; mc.term = true; }
  else { mc.i = idx - 1; mc.step = -1; mc.term = false; WHY(5); }    // back along that walk's path (its seed included)
  WHY(7);
#undef WHY
  return true;
}

// RESUME: the walks of the list were started by the thread kernel this round (promo_list); go on where they stand.
template <bool RESUME>
__global__ __launch_bounds__(64) void ext_walk_long_kernel(WalkArgs A, const uint32_t* __restrict__ long_list, uint64_t n_walks,
                                                           const unsigned long long* __restrict__ list_count) {
  const unsigned long long n_list = RESUME ? *list_count : (unsigned long long)gridDim.x;
  for (unsigned long long li = blockIdx.x; li < n_list; li += gridDim.x) {
  const uint32_t r = long_list[li];
  const int lane = threadIdx.x;
  const uint32_t o = A.order[r];
  uint32_t ns = 0, nr_new = 0;
  uint64_t tot;
  int dir0 = 0;
  uint32_t cur0 = o;
  u64 seen = UNCLAIMED64;                     // per lane: what its last claim found, looked at when it claims again (note_claim)
  if (RESUME) {
    const uint32_t info = A.res_info[r];
    dir0 = (int)(info >> 31);
    ns = info & 0x7FFFFFFFu;
    cur0 = A.res_cur[r];
    nr_new = dir0 ? A.nr_out[r] : 0;
    tot = A.totw_out[r];
  } else {
    const bool isvoid = RANK(A.claim_old[o]) < r || RANK(__hip_atomic_load(&A.claim[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r;
    if (isvoid) {                                 // seed currently traversed; the memo stays for later
      if (lane == 0) { A.nr_out[r] = UNCLAIMED; A.nl_out[r] = 0; A.totw_out[r] = 0; }
      continue;
    }
    tot = A.weight[o];
    if (lane == 0) seen = claim_node(A, o, r, 0);
  }
  const uint32_t ns_start = ns;
  uint32_t nseq = 0;                          // sequential steps (debug statistics)
  WhyStat why;
  for (int dir = dir0; dir < 2; dir++) {
    const RowView adj = dir == 0 ? A.adjR : A.adjL;
    uint32_t cur = (RESUME && dir == dir0) ? cur0 : o;
    MemoCursor mc;
    // own memo first (its steps of this direction), else whatever memo the k1-mer was last written into
    bool following;
    if (RESUME && dir == dir0) following = memo_follow(A, A.hint[cur], cur, dir, mc);
    else {
      const uint32_t own = A.mvalid[r] ? (uint32_t)(((A.moff[r] + 2) << 2) | HINT_SEED) : NOHINT;
      following = memo_follow(A, own, o, dir, mc) || memo_follow(A, A.hint[o], o, dir, mc);
    }
    uint32_t cool = 0;                        // sequential steps to take before trusting a memo again
    Adj4 cand = {{-1, -1, -1, -1}};           // row of `cur` while walking sequentially
    bool have_cand = false;
    while (true) {
      if (following) {
        // 64 memo steps per trip.  A word with the top bit set (mark, header, hole) ends the segment; going backwards
        // the segment also ends after the memo walk's seed (the word before a seed is its header).
        const int64_t pi = mc.i + (int64_t)lane * mc.step;  // pool index of this lane's step
        const uint32_t mine = A.pool[pi];
        const u64 stopm = __ballot((mine & MEMO_HI) != 0);
        const uint32_t nchunk = stopm ? (uint32_t)(__ffsll((long long)stopm) - 1) : 64u;   // steps before the first stop word
        const bool at_mark = nchunk < 64u && __shfl(mine, (int)nchunk, 64) == MEMO_MARK;
        const bool is_term = mc.term && at_mark && (uint32_t)lane == nchunk;    // the memo walk stopped here: would I?
        const bool checked = (uint32_t)lane < nchunk || is_term;
        bool ok = false;
        if (checked) {
          const uint32_t before = lane == 0 ? cur : A.pool[pi - mc.step];
          // an entry counts only if its k1-mer's hint points at exactly this pool word (validated entries are distinct)
          bool valid = is_term || (A.hint[mine] >> 2) == (uint32_t)pi;
          if (valid) {
            Adj4 cd = adj[before];
            uint32_t bw;
            int b = decide(cd, r, A.claim, A.claim_old, A.weight, o, bw);
            uint32_t chosen = b < 0 ? NONE32 : (uint32_t)adj_get(cd, b);
            ok = chosen == (is_term ? NONE32 : mine);
          }
        }
        const u64 bad = __ballot(checked && !ok);
        const uint32_t m = bad ? (uint32_t)(__ffsll((long long)bad) - 1) : 64u;
        const uint32_t conf = min(m, nchunk);               // confirmed memo steps: lanes [0, conf)
        uint64_t myw = 0;
        if ((uint32_t)lane < conf) {
          note_claim(A, seen, r);
          seen = claim_node(A, mine, r, ns + lane + 1);
          myw = A.weight[mine];
        }
        for (int off = 32; off > 0; off >>= 1) myw += __shfl_xor(myw, off, 64);
        tot += myw;
        if (conf > 0) cur = __shfl(mine, (int)conf - 1, 64);
        if (A.dbg && lane == 0 && conf) atomicAdd(&A.dbg[1], (unsigned long long)conf);
        ns += conf;
        mc.i += (int64_t)conf * mc.step;
        if (m == 64u) {
          if (nchunk < 64u) {
            if (is_term_any(mc.term, at_mark)) break;      // the terminal lane agreed: the walk ends where the memo's walk ended
            following = false;                // the segment just runs out (hole, or followed backwards): go on from its last k1-mer
            have_cand = false;
          }
          continue;
        }
        following = false;                    // decision m changed (or its entry is not trustworthy): go on sequentially
        have_cand = false;
        if (A.dbg && lane == 0) atomicAdd(&A.dbg[conf == 0 ? 8 : 9], 1ULL);
        if (conf == 0) cool = 2;
        continue;
      }
      // sequential step from `cur`: lanes 0..3 each fetch one candidate (claims, weight, hint and its own row, one
      // memory round trip), the decision is made by everybody from the shuffled weights
      if (!have_cand) cand = adj[cur];
      const int myc = lane == 0 ? cand.v[0] : lane == 1 ? cand.v[1] : lane == 2 ? cand.v[2] : lane == 3 ? cand.v[3] : -1;
      uint32_t hmy = NOHINT;
      uint32_t wmy = 0;
      Adj4 row = {{-1, -1, -1, -1}};
      bool avail = false;
      if (myc >= 0) {
        u64 cl = __hip_atomic_load(&A.claim[myc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64 co = A.claim_old[myc];
        hmy = A.hint[myc];
        wmy = A.weight[(uint32_t)myc];
        row = adj[myc];
        avail = RANK(cl) > r && RANK(co) >= r;
      }
      const uint32_t am = (uint32_t)(__ballot(avail) & 0xFull);
      if (!am) break;
      const uint32_t w0 = __shfl(wmy, 0, 64), w1 = __shfl(wmy, 1, 64), w2 = __shfl(wmy, 2, 64), w3 = __shfl(wmy, 3, 64);
      int best = -1;
      uint32_t bw = 0;
#define CONSIDER(b, wb) if ((am >> b) & 1u) { if (best < 0 || wb > bw) { best = b; bw = wb; } }
      CONSIDER(0, w0) CONSIDER(2, w2) CONSIDER(1, w1) CONSIDER(3, w3)
#undef CONSIDER
      const uint32_t taken = (uint32_t)__shfl(myc, best, 64);
      const uint32_t hh = (uint32_t)__shfl((int)hmy, best, 64);
      cand.v[0] = __shfl(row.v[0], best, 64); cand.v[1] = __shfl(row.v[1], best, 64);
      cand.v[2] = __shfl(row.v[2], best, 64); cand.v[3] = __shfl(row.v[3], best, 64);
      have_cand = true;
      if (lane == 0) { note_claim(A, seen, r); seen = claim_node(A, taken, r, ns + 1); }
      tot += bw;
      ns++;
      nseq++;
      cur = taken;
      if (cool) cool--;
      else following = memo_follow(A, hh, taken, dir, mc, &why);
    }
    if (dir == 0) nr_new = ns;
  }
  note_claim(A, seen, r);
  if (A.dbg && lane == 0) {
    atomicMax(&A.dbg[10], ((unsigned long long)nseq << 48) | ((unsigned long long)min(why.c2, 4095u) << 36) |
                              ((unsigned long long)min(why.c4, 4095u) << 12) | (unsigned long long)min(why.c7, 4095u));
    atomicMax(&A.dbg[11], (unsigned long long)(ns - ns_start));
  }
  if (lane == 0) {
    A.nr_out[r] = nr_new;
    A.nl_out[r] = ns - nr_new;
    A.totw_out[r] = tot;
    const uint32_t mine = ns - ns_start;
    if (mine) atomicAdd(&A.wave_steps_counter[blockIdx.x & 63], (unsigned long long)mine);   // 64 slots: no single hot address
  }
  }
}

// ---- fixpoint audit, after the last block settles (always on; one pass over the claims, no walking): the claims are
// the fixpoint iff every claimed k1-mer is what its walk's greedy rule picks at the step before it, every walk ends
// where its rule finds nothing, and every walk owns exactly its recorded steps.  For walk r at step p a k1-mer is
// traversed if a lower rank owns it or r owns it at a step <= p.  A walk that fails is made dirty and the rounds go
// on with every block reopened -- the rounds' change tracking is an optimisation, this is the definition.
__global__ void ext_audit_nodes_kernel(const u64* __restrict__ claim, uint64_t n2, const RowView adjR, const RowView adjL,
                                       const WordView weight, const uint32_t* __restrict__ order,
                                       const uint32_t* __restrict__ nr_a, const uint32_t* __restrict__ nl_a, uint64_t ns,
                                       uint32_t* __restrict__ owned, uint8_t* __restrict__ dirty, unsigned long long* __restrict__ counters) {
  for (uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; y < n2; y += (uint64_t)gridDim.x * blockDim.x) {
    const u64 c = claim[y];
    const uint32_t r = RANK(c), p = POS(c);
    if (r == UNCLAIMED) continue;
    if (r >= ns) { atomicAdd(&counters[0], 1ULL); continue; }
    atomicAdd(&owned[r], 1u);
    const uint32_t nr = nr_a[r], nl = nl_a[r];
    bool bad = false;
    if (nr == UNCLAIMED || p > nr + nl) bad = true;                     // claim of a void walk / beyond its record
    else {
      if (p == 0) bad = order[r] != (uint32_t)y;
      else {
        // the step before: position p-1 going right, and the seed again for the first step to the left
        const bool right = p <= nr;
        const u64 want = CLAIM(r, p == nr + 1 ? 0u : p - 1);
        const Adj4 back = right ? adjL[y] : adjR[y];
        int32_t x = -1;
#pragma unroll
        for (int q = 0; q < 4; q++) if (back.v[q] >= 0 && claim[back.v[q]] == want) x = back.v[q];
        if (x < 0) bad = true;
        else {
          const Adj4 cd = right ? adjR[x] : adjL[x];
          const int b = audit_decide(cd, r, p - 1, claim, weight);
          bad = b < 0 || adj_get(cd, b) != (int32_t)y;
        }
      }
      if (!bad && p == nr) bad = audit_decide(adjR[y], r, nr, claim, weight) >= 0;                      // right end
      if (!bad && (nl ? p == nr + nl : p == 0)) bad = audit_decide(adjL[y], r, nr + nl, claim, weight) >= 0;   // left end
    }
    if (bad) { dirty[r] = 1; atomicAdd(&counters[0], 1ULL); }
  }
}

__global__ void ext_audit_walks_kernel(const u64* __restrict__ claim, const uint32_t* __restrict__ order, const uint32_t* __restrict__ nr_a,
                                       const uint32_t* __restrict__ nl_a, uint64_t ns, const uint32_t* __restrict__ owned,
                                       uint8_t* __restrict__ dirty, unsigned long long* __restrict__ counters) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ns) return;
  const u64 cs = claim[order[r]];
  const uint32_t nr = nr_a[r];
  const bool bad = nr == UNCLAIMED ? !(RANK(cs) < r && owned[r] == 0) : (cs != CLAIM((uint32_t)r, 0) || owned[r] != nr + nl_a[r] + 1u);
  if (bad) { dirty[r] = 1; atomicAdd(&counters[1], 1ULL); }
}

// ---- audit (SHN_EXT_AUDIT=1, tests and stress runs): re-derive every walk from the converged claims alone, one thread
// per walk.  For walk r a k1-mer is traversed if a lower rank owns it or r owns it at a position already passed; the
// greedy choice at every step must be the k1-mer r owns at the next position, and the walk must end where its
// recorded counts say.  counters: [0] walks that disagree [1] the lowest such rank
__global__ void ext_audit_kernel(WalkArgs A, uint64_t ns, unsigned long long* __restrict__ counters) {
  uint64_t r64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r64 >= ns) return;
  const uint32_t r = (uint32_t)r64;
  const uint32_t o = A.order[r];
  const u64 cs = A.claim[o];
  const uint32_t nr = A.nr_out[r], nl = A.nl_out[r];
  bool bad = false;
  if (nr == UNCLAIMED) bad = !(RANK(cs) < r);
  else if (cs != CLAIM(r, 0)) bad = true;
  else {
    uint32_t pos = 0;
    uint64_t tot = A.weight[o];
    for (int dir = 0; dir < 2 && !bad; dir++) {
      const RowView adj = dir == 0 ? A.adjR : A.adjL;
      const uint32_t end = dir == 0 ? nr : nr + nl;
      uint32_t cur = o;
      while (true) {
        Adj4 cd = adj[cur];
        int best = -1;
        uint32_t bw = 0;
#pragma unroll
        for (int bi = 0; bi < 4; bi++) {
          const int b = bi == 0 ? 0 : bi == 1 ? 2 : bi == 2 ? 1 : 3;
          if (cd.v[b] < 0) continue;
          const u64 c = A.claim[cd.v[b]];
          const bool avail = RANK(c) > r || (RANK(c) == r && POS(c) > pos);
          const uint32_t w = A.weight[(uint32_t)cd.v[b]];
          if (avail && (best < 0 || w > bw)) { best = b; bw = w; }
        }
        if (best < 0) { if (pos != end) bad = true; break; }
        if (pos == end) { bad = true; break; }                // the recorded walk stopped, the rule goes on
        const uint32_t nx = (uint32_t)adj_get(cd, best);
        if (A.claim[nx] != CLAIM(r, pos + 1)) { bad = true; break; }
        pos++; tot += bw; cur = nx;
      }
    }
    if (!bad && tot != A.totw_out[r]) bad = true;
  }
  if (bad) { atomicAdd(&counters[0], 1ULL); atomicMin(&counters[1], (unsigned long long)r); }
}

// ---- Seeds that cannot survive, decided from the graph alone (round 6).  Let p be a left neighbour of the seed s whose ONLY right
// neighbour is s (a forced link p -> s), and let some walk of rank below rank(s) traverse p -- p's own, if p is a seed of lower rank.
// Whatever walk W gets to p first (extension_correction.py:223-245): it is seeded on p and extends right first -- its one candidate
// is s; or it arrives at p extending right -- the same; or it arrives at p extending left, which it can only do from p's one right
// neighbour, s.  In every case s is traversed by W or by an earlier walk, W's rank is at most rank(p) < rank(s), and the walk seeded
// on s is void (:346).  By induction along a chain of forced links p_d -> ... -> p_1 -> s a lower rank ANYWHERE on the chain is
// enough (whoever traverses p_i traverses p_{i-1} or came from it), and the mirror image holds on the right (q's only left
// neighbour is s).  So of the seeds of a stretch without branches only the local rank minima can start a surviving walk: at
// BASELINE configs[2] 98.6 % of the 186 M walks end void, most of them swallowed by a neighbour on their own unitig after a few
// steps -- each cost its seed claim and the lines of its first candidates, and robbed whoever it met.  Here such a seed costs the
// records of its chain neighbours (one 64-byte line per hop) and is never launched.  `hops` links are followed on each side; from
// the second hop on only through k1-mers with a single neighbour on that side.
__device__ __forceinline__ int adj_count(const Adj4& a) { return (a.v[0] >= 0) + (a.v[1] >= 0) + (a.v[2] >= 0) + (a.v[3] >= 0); }
__device__ __forceinline__ bool ext_chain_has_lower(const Rec* __restrict__ rec, uint32_t o, uint32_t r, uint32_t hops) {
  const Adj4 L0 = *(const Adj4*)((const char*)&rec[o] + 16), R0 = *(const Adj4*)((const char*)&rec[o]);
#pragma unroll
  for (int side = 0; side < 2; side++) {                       // 0: left neighbours (forced = their right row has one entry), 1: right
    Adj4 nb = side == 0 ? L0 : R0;
    for (uint32_t h = 0; h < hops; h++) {
      const int cnt = adj_count(nb);
      if (cnt == 0 || (h > 0 && cnt != 1)) break;
      Adj4 next; next.v[0] = next.v[1] = next.v[2] = next.v[3] = -1;
      bool forced_one = false;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        if (nb.v[b] < 0) continue;
        const Rec* q = &rec[(uint32_t)nb.v[b]];
        const Adj4 toward = side == 0 ? *(const Adj4*)((const char*)q) : *(const Adj4*)((const char*)q + 16);     // the row that points back at us
        if (adj_count(toward) != 1) continue;                  // not forced: a walk may pass it without coming our way
        if (q->seed_rank < r) return true;
        if (cnt == 1) { next = side == 0 ? *(const Adj4*)((const char*)q + 16) : *(const Adj4*)((const char*)q); forced_one = true; }
      }
      if (!forced_one) break;
      nb = next;
    }
  }
  return false;
}

// classify the dirty walks of the open block: long ones (memo or recorded length) go to the wavefront kernel,
// the others to the thread kernel.  counters: [0] long [1] dirty walks that hold claims and have no current memo [2] short [3] dirty walks
__global__ __launch_bounds__(1024) void ext_plan_kernel(uint32_t* nr, uint32_t* nl, uint64_t ns, uint32_t frozen,
                                const uint8_t* __restrict__ mvalid, const uint32_t* __restrict__ mR, const uint32_t* __restrict__ mL,
                                const uint8_t* __restrict__ dirty, uint32_t* __restrict__ long_list, uint32_t* __restrict__ short_list,
                                unsigned long long* __restrict__ counters, uint32_t long_walk, uint8_t* __restrict__ coarse,
                                const u64* __restrict__ fresh_claim, const uint32_t* __restrict__ order, uint64_t* __restrict__ totw,
                                const Rec* __restrict__ rec, uint32_t settle_hops, uint8_t* __restrict__ settled,
                                unsigned long long* __restrict__ n_settled, uint8_t* __restrict__ robsat = nullptr,
                                unsigned long long* __restrict__ n_robsat = nullptr, const uint32_t* __restrict__ log_head = nullptr,
                                unsigned long long* __restrict__ rel_steps = nullptr) {
  // ns here = current rank limit (walks >= limit have not started yet); walks < frozen are final and never run
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + frozen;
  const bool isd_all = r < ns && dirty[r];
  // coarse[i]: one of the 64 walks frozen + 64 i .. is dirty (the begin pass asks it before dirty[]: 1/64 of the bytes, they stay
  // in the L2 while the claims stream past)
  { const unsigned long long anyd = __ballot(isd_all); if ((threadIdx.x & 63) == 0) coarse[(r - frozen) >> 6] = anyd ? 1 : 0; }
  // fresh_claim != NULL: the first round of a block that has just opened -- every claim there is belongs to a final walk, so a walk
  // whose seed is claimed (by a lower rank: final walks are all lower) is void for good: its record is written here and it is on
  // no list.  At BASELINE configs[2] 98.6 % of the walks are void, most of them through walks of EARLIER blocks: the walk kernel of
  // the second and third block then runs over the survivors, packed -- not one live walk among 63 lanes that look at their seed and
  // idle until the wavefront's longest walk ends.
  bool isd = isd_all;
  if (isd && settled[r]) { nr[r] = UNCLAIMED; nl[r] = 0; totw[r] = 0; isd = false; }           // void for good (below): it never runs
  else if (isd && fresh_claim && RANK(fresh_claim[order[r]]) < r) { nr[r] = UNCLAIMED; nl[r] = 0; totw[r] = 0; isd = false; }
  else if (isd && settle_hops && ext_chain_has_lower(rec, order[r], (uint32_t)r, settle_hops)) {
    settled[r] = 1; nr[r] = UNCLAIMED; nl[r] = 0; totw[r] = 0; isd = false;
    atomicAdd(n_settled, 1ULL);
  }
  bool lg = false;
  if (isd) {
    uint32_t a = nr[r];
    uint32_t len = a == UNCLAIMED ? 0 : a + nl[r];
    if (mvalid[r]) len = max(len, mR[r] + mL[r]);
    lg = len >= long_walk;
  }
  // one atomic per block of 1024 and list (one per wavefront on three single addresses was 39 us per launch)
  __shared__ uint32_t wl[16], wsh[16];
  __shared__ unsigned long long bl, bs, bd, bh;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long below = (1ULL << lane) - 1ULL;
  const unsigned long long lm = __ballot(isd && lg), sm = __ballot(isd && !lg);
  if (lane == 0) { wl[wid] = (uint32_t)__popcll(lm); wsh[wid] = (uint32_t)__popcll(sm); }
  if (threadIdx.x == 0) { bd = 0; bh = 0; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tl = 0, ts = 0;
    const int nw = (int)(blockDim.x >> 6);
    for (int w = 0; w < nw; w++) { tl += wl[w]; ts += wsh[w]; }
    bl = tl ? atomicAdd(&counters[0], (unsigned long long)tl) : 0ULL;
    bs = ts ? atomicAdd(&counters[2], (unsigned long long)ts) : 0ULL;
  }
  {                                                                // dirty walks, the ones settled above included (they "ran")
    const unsigned long long dm = __ballot(isd_all);
    if (lane == 0 && dm) atomicAdd(&bd, (unsigned long long)__popcll(dm));
    // ... of them the ones that hold claims (a record of a live walk): with none, the round's begin pass has nothing to release
    // (counted: the holders WITHOUT a current memo -- with none of those, the dirty walks release their claims themselves, from
    // their memos: ext_release_memo_kernel)
    // (... nor a claim log: a walk that last ran in a bulk round gives its claims back through the log it wrote there)
    const bool holds = isd_all && nr[r] != UNCLAIMED;
    const bool has_log = holds && mvalid[r] != 2 && log_head && log_head[r] != LOG_LOST;      // (NONE32: it took no step -- its seed is all it holds)
    const unsigned long long hm = __ballot(holds && mvalid[r] != 2 && !has_log);
    if (lane == 0 && hm) atomicAdd(&bh, (unsigned long long)__popcll(hm));
    if (rel_steps && __ballot(holds)) {                           // the claims a targeted release would have to visit (most wavefronts hold no dirty walk at all)
      unsigned long long st = holds ? (unsigned long long)nr[r] + nl[r] + 1ULL : 0ULL;
      for (int o = 32; o > 0; o >>= 1) st += __shfl_down(st, o, 64);
      if (lane == 0 && st) atomicAdd(rel_steps, st);
    }
    if (robsat) {      // (development, SHN_EXT_XTIME: of those, the walks that were robbed while they sat out -- their chain of claims has a gap)
      const unsigned long long rm = __ballot(isd_all && nr[r] != UNCLAIMED && mvalid[r] != 2 && robsat[r]);
      if (lane == 0 && rm) atomicAdd(n_robsat, (unsigned long long)__popcll(rm));
      if (isd_all && nr[r] != UNCLAIMED && mvalid[r] != 2) { const unsigned long long len = (unsigned long long)nr[r] + nl[r]; atomicMax(n_robsat + 1, len); atomicAdd(n_robsat + 2, len); }
      if (isd_all) robsat[r] = 0;                                    // (it runs now: what it holds afterwards is a fresh chain)
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && bd) atomicAdd(&counters[3], bd);
  if (threadIdx.x == 0 && bh) atomicAdd(&counters[1], bh);
  __syncthreads();
  uint32_t ol = 0, os = 0;
  for (int w = 0; w < wid; w++) { ol += wl[w]; os += wsh[w]; }
  if (isd && lg) long_list[bl + ol + __popcll(lm & below)] = (uint32_t)r;
  if (isd && !lg) short_list[bs + os + __popcll(sm & below)] = (uint32_t)r;
}

// after the walkers: every walk that ran alive and is long enough gets a NEW memo slot for its new path (filled from
// the claims by ext_mark_kernel); old slots stay as they are -- the hints of k1-mers the walk no longer owns still lead
// to an intact path.  When the pool is full, no more memos are made (they only save time).
__global__ void ext_memo_plan_kernel(uint8_t* __restrict__ ran, uint8_t* __restrict__ dirty,
                                     const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, const uint32_t* __restrict__ order,
                                     uint32_t frozen, uint32_t limit, uint64_t* __restrict__ moff, uint32_t* __restrict__ mR,
                                     uint32_t* __restrict__ mL, uint8_t* __restrict__ mvalid, uint8_t* __restrict__ fill,
                                     uint32_t* __restrict__ pool, unsigned long long* __restrict__ cursor, uint64_t pool_cap, uint32_t memo_min,
                                     uint32_t* __restrict__ log_head = nullptr) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + frozen;
  if (r >= limit) return;
  const uint8_t did_run = dirty[r];              // end-of-round bookkeeping: who ran, clean slate for the marks
  ran[r] = did_run;
  dirty[r] = 0;
  // (a walk that ran in a round without logging -- a round that makes memos -- holds other claims than its log says; one that took
  // no step holds its seed and nothing else: the empty log says so -- such a walk gets no memo slot either)
  if (log_head && did_run && memo_min != 0xFFFFFFFFu) log_head[r] = (nr[r] != UNCLAIMED && nr[r] + nl[r] == 0) ? NONE32 : LOG_LOST;
  uint8_t f = 0;
  if (did_run && nr[r] != UNCLAIMED) {
    const uint32_t R = nr[r], L = nl[r];
    if (R + L >= memo_min) {
      const uint64_t cap = (uint64_t)R + L + 5;
      const unsigned long long off = atomicAdd(cursor, (unsigned long long)cap);
      if (off + cap + 64 <= pool_cap) {
        pool[off] = MEMO_MARK; pool[off + 1] = R | MEMO_HI; pool[off + 2] = order[r];
        pool[off + 3 + R] = MEMO_MARK; pool[off + 4 + R + L] = MEMO_MARK;
        moff[r] = off; mR[r] = R; mL[r] = L; mvalid[r] = 2; f = 1;     // 2: the memo holds exactly the walk's claims (until it runs again)
      }
    }
  }
  if (did_run && !f && mvalid[r]) mvalid[r] = 1;     // ran without a new slot (a bulk round, a full pool, void now): the old memo is a hint only
  fill[r] = f;
}

// The release of a round that re-runs few walks, all of which have a current memo (the path of their last run, written from the
// claims by the mark pass of the round they ran in; they have not run since): a wavefront per dirty walk goes through its memo and
// gives back what the walk still owns.  The begin pass below streams all 11.6 GB of claims of BASELINE configs[2] to find those
// few thousand k1-mers -- 2.7 ms a round, half of the rounds of a step re-run fewer than 10,000 walks.
__global__ __launch_bounds__(64) void ext_release_memo_kernel(const uint32_t* __restrict__ long_list, uint64_t n_long, const uint32_t* __restrict__ short_list,
                                                              uint64_t n_short, const uint32_t* __restrict__ nr, const uint64_t* __restrict__ moff,
                                                              const uint32_t* __restrict__ mR, const uint32_t* __restrict__ mL, const uint32_t* __restrict__ pool,
                                                              u64* claim, uint8_t* __restrict__ chunk, const uint8_t* __restrict__ mvalid = nullptr,
                                                              const uint32_t* __restrict__ order = nullptr, const uint32_t* __restrict__ logpool = nullptr,
                                                              const uint32_t* __restrict__ log_head = nullptr, const uint8_t* __restrict__ log_cnt = nullptr) {
  const uint64_t idx = blockIdx.x;
  if (idx >= n_long + n_short) return;
  const uint32_t r = idx < n_long ? long_list[idx] : short_list[idx - n_long];
  if (nr[r] == UNCLAIMED) return;                                        // void: it holds nothing
  if (mvalid && mvalid[r] != 2) {
    // no current memo: the walk last ran in a bulk round and wrote a claim log there -- its seed, then the chunks from the last one
    // back; a k1-mer of the log that is no longer the walk's (a lower rank took it) stays as it is
    auto give_back = [&](uint32_t node) {
      const u64 c = __hip_atomic_load(&claim[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (RANK(c) == r && atomicCAS(&claim[node], c, UNCLAIMED64) == c && chunk) chunk[node >> CHUNK_SHIFT] = 1;
    };
    if (threadIdx.x == 0) give_back(order[r]);
    uint32_t ch = log_head[r], cnt = log_cnt[r];
    while (ch != NONE32 && ch != LOG_LOST) {
      const uint32_t* c = logpool + (uint64_t)ch * LOG_WORDS;
      if (threadIdx.x >= 1 && threadIdx.x <= cnt) give_back(c[threadIdx.x]);
      ch = c[0];
      cnt = LOG_PER;
    }
    return;
  }
  const uint64_t off = moff[r];
  const uint32_t R = mR[r], L = mL[r];
  for (uint32_t j = threadIdx.x; j <= R + L; j += 64) {
    const uint32_t node = pool[off + (j <= R ? 2 : 3) + j];
    if (node & 0x80000000u) continue;                                    // a step that was robbed before the memo was written
    if (atomicCAS(&claim[node], CLAIM(r, j), UNCLAIMED64) == CLAIM(r, j) && chunk) chunk[node >> CHUNK_SHIFT] = 1;
  }
}

// start of a round: snapshot the claims, drop the claims of the walks that are about to re-run, clear the round's counters
// (copy == 0: the snapshot is already the claims -- ext_mark_kernel brought it up to date where the last round changed something)
__global__ void ext_round_begin_kernel(u64* __restrict__ claim, u64* __restrict__ snap, uint64_t n2, const uint8_t* __restrict__ dirty,
                                       uint64_t ns, unsigned long long* __restrict__ d_cnt, int copy, uint8_t* __restrict__ chunk,
                                       uint32_t frozen, uint32_t limit, const uint8_t* __restrict__ coarse) {
  // four claims per thread, as two 16-byte loads (n2 is padded to a multiple of 4 by the allocation; the claims are 16-byte aligned):
  // with one 8-byte load per thread the pass ran at 3 TB/s
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) { d_cnt[6] = 0; d_cnt[7] = 0; d_cnt[13] = 0; d_cnt[14] = 0; }      // changed k1-mers, (spare), walks handed over, the head of their list
  const uint64_t o0 = t * 4;
  if (o0 >= n2) return;
  ulonglong2 c01 = ((const ulonglong2*)(claim + o0))[0], c23 = ((const ulonglong2*)(claim + o0))[1];
  if (copy) { ((ulonglong2*)(snap + o0))[0] = c01; ((ulonglong2*)(snap + o0))[1] = c23; }
  // (only walks of the open block can be dirty: the flag of a final walk's k1-mer -- most claimed k1-mers in the later blocks --
  // is not looked up: a random byte read per claimed k1-mer otherwise)
  u64 c[4] = {c01.x, c01.y, c23.x, c23.y};
  bool any = false;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t rk = RANK(c[q]);
    if (o0 + q < n2 && rk >= frozen && rk < limit && rk < ns && coarse[(rk - frozen) >> 6] && dirty[rk]) { c[q] = UNCLAIMED64; any = true; }
  }
  if (any) {
    ((ulonglong2*)(claim + o0))[0] = ulonglong2{c[0], c[1]}; ((ulonglong2*)(claim + o0))[1] = ulonglong2{c[2], c[3]};
    if (chunk) chunk[o0 >> CHUNK_SHIFT] = 1;
  }
}

// after a round: every k1-mer whose owner changed dirties the walks that looked at it; the k1-mers of the walks
// that got a memo slot are written into it (memo + hint)
__global__ void ext_mark_kernel(const u64* __restrict__ claim, u64* claim_old, uint64_t n2, Rec* __restrict__ rec,
                                uint8_t* __restrict__ dirty, const uint8_t* __restrict__ ran,
                                unsigned long long* __restrict__ n_changed, uint32_t frozen, uint32_t limit,
                                const uint8_t* __restrict__ fill /* NULL: no walk got a memo slot this round */, const uint64_t* __restrict__ moff, const uint32_t* __restrict__ mR,
                                uint32_t* __restrict__ pool,
                                const uint32_t* __restrict__ nr_a, const uint32_t* __restrict__ nl_a, int precise,
                                const uint8_t* __restrict__ chunk, uint8_t* __restrict__ robsat = nullptr) {
  const RowView adjR = rows_R(rec), adjL = rows_L(rec);
  const WordView weight = words_weight(rec);
  // grid-stride: the change counter costs one atomic per block (one per wavefront on a single address was the
  // most expensive thing in this kernel).  A wavefront takes 64 k1-mers at a time and, in rounds that re-run few walks
  // (chunk != NULL), looks only at the 128-byte lines of claims (16 k1-mers, one flag) a claim was written in this round
  // (claim_node and the release of the begin pass flag them): where nothing was written nothing changed, and every k1-mer of a
  // walk that ran was written.  The k1-mers of a walk are scattered over the table, so the flags have to be this fine: with a flag
  // per 64 k1-mers a round of 500 k short walks read 3 GB of claims and snapshot for its 3 M written k1-mers.
  uint32_t my_changed = 0;
  const uint64_t n_groups = (n2 + 63) >> 6;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  // the part every written k1-mer goes through; true: its owner changed to a higher rank (or nobody) -- the walks around it have to be looked at (mark_around)
  auto mark_node = [&](const uint64_t y, uint32_t& a_out) -> bool {
    const u64 oy = claim_old[y];
    const uint32_t a = RANK(oy);
    const u64 cy = claim[y];
    const uint32_t b = RANK(cy);
    // precise marks read the snapshot only at y itself: bring it up to date here, and the next round's begin pass has nothing
    // to copy (8 of the 32 bytes the two passes move per k1-mer and round)
    if (precise && oy != cy) claim_old[y] = cy;
    // (only walks of the open block can have run; a walk that got a slot ran.  In a bulk round nobody gets one, and a claimed
    // k1-mer whose owner did not change costs nothing beyond the two streams)
    if (fill && b >= frozen && b < limit && fill[b]) {            // slot layout: see MemoCursor
      const uint32_t pos = POS(cy), R = mR[b];
      const uint64_t idx = moff[b] + (pos <= R ? 2 : 3) + pos;
      if (pos) pool[idx] = (uint32_t)y;
      rec[y].hint = (uint32_t)(idx << 2) | (pos == 0 ? HINT_SEED : pos <= R ? HINT_R : HINT_L);
    }
    if (a == b) return false;
    my_changed++;
    // Who has to look again?  Walk x treats y as traversed iff its owner's rank is below x, and removing a
    // candidate it did not choose never changes a greedy choice -- so only walks for which y BECAME available
    // (a < x < b) are affected: the walks that stood next to y (owners of its 8 neighbours) and the walk seeded
    // on y (void while y belonged to a lower rank).  The old owner re-runs if it was robbed while it sat out this
    // round (one that ran this round gave y up knowingly; one that lost it during its run is caught by the verify
    // kernel); the new owner ran this round.  Walks below `frozen` are final, walks at or above `limit` have not
    // started (they all run when their phase opens).
#define MARK(x) if ((x) >= frozen && (x) < limit) dirty[x] = 1
    if (a != UNCLAIMED && !ran[a]) { MARK(a); if (robsat && a >= frozen && a < limit) robsat[a] = 1; }
    a_out = a;
    return b >= a;
  };
  auto mark_around = [&](const uint64_t y, const uint32_t a) {
    const uint32_t b = RANK(claim[y]);
#define MARKX(x) if (a < (x) && (x) < b) MARK(x)
    const uint32_t sr = rec[y].seed_rank;                        // (the same line as the two rows)
    MARKX(sr);
    Adj4 L = adjL[y], R = adjR[y];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      int32_t nb = q < 4 ? L.v[q] : R.v[q - 4];
      if (nb < 0) continue;
      const u64 cz = claim[nb];
      const uint32_t z = RANK(cz);
      if (!precise) { uint32_t x = RANK(claim_old[nb]); MARKX(x); MARKX(z); continue; }
      // Precise: y became available to the walk z that stands next to it (a < z < b).  z has to look again only if its greedy
      // choice at nb, re-made from the final claims, is not the step it recorded there -- the same test the fixpoint audit
      // applies to every k1-mer, here applied where something changed.  (A former owner of nb either re-ran or was robbed
      // while it sat out and is marked at nb itself.)
      if (!(a < z && z < b) || z < frozen || z >= limit || dirty[z]) continue;
      const uint32_t pos = POS(cz), nrz = nr_a[z], nlz = nl_a[z];
      if (nrz == UNCLAIMED || pos > nrz + nlz) { dirty[z] = 1; continue; }
      const bool dirR = q < 4;                                   // y is a right candidate of nb (nb is a left neighbour of y)
      if (dirR ? pos > nrz : (pos != 0 && pos <= nrz)) continue;  // z left nb in the other direction: it never looked at y from here
      const uint32_t thr = dirR ? pos : (pos == 0 ? nrz : pos);   // own steps up to here count as traversed
      const bool has_next = dirR ? pos < nrz : (pos == 0 ? nlz > 0 : pos < nrz + nlz);
      const uint32_t next_pos = dirR ? pos + 1 : (pos == 0 ? nrz + 1 : pos + 1);
      const Adj4 cd = dirR ? adjR[nb] : adjL[nb];
      const int bsel = audit_decide(cd, z, thr, claim, weight);
      const bool same = has_next ? (bsel >= 0 && claim[adj_get(cd, bsel)] == CLAIM(z, next_pos)) : bsel < 0;
      if (!same) dirty[z] = 1;
    }
#undef MARKX
#undef MARK
  };
  if (!chunk) {
    for (uint64_t ch = wave; ch < n_groups; ch += n_waves) {
      const uint64_t y = (ch << 6) + lane;
      uint32_t a;
      if (y < n2 && mark_node(y, a)) mark_around(y, a);
    }
  } else {
    // 64 groups of 64 k1-mers per trip: every lane fetches the four line flags of one group, the wavefront then visits the groups
    // that have one set.  (Putting the k1-mers whose surroundings have to be looked at on a list for a second launch, 64 to a
    // wavefront, was slower: 235 against 210 ms per step -- the pass is bound by its random accesses, not by their latency.)
    static_assert(CHUNK_SHIFT == 4, "four line flags per group of 64 k1-mers");
    const uint64_t n_super = (n_groups + 63) >> 6;
    for (uint64_t sg = wave; sg < n_super; sg += n_waves) {
      const uint64_t g = (sg << 6) + lane;
      const uint32_t fl = g < n_groups ? *(const uint32_t*)(chunk + g * 4) : 0u;
      unsigned long long todo = __ballot(fl != 0);
      while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t flj = (uint32_t)__shfl((int)fl, j, 64);
        const uint64_t y = ((((sg << 6) + (uint64_t)j)) << 6) + lane;
        uint32_t a = 0;
        if (y < n2 && ((flj >> (8 * (lane >> 4))) & 0xFFu) && mark_node(y, a)) mark_around(y, a);
      }
    }
  }
  __shared__ unsigned long long blk_changed;
  if (threadIdx.x == 0) blk_changed = 0;
  __syncthreads();
  if (my_changed) atomicAdd(&blk_changed, (unsigned long long)my_changed);
  __syncthreads();
  if (threadIdx.x == 0 && blk_changed) atomicAdd(n_changed, blk_changed);
}

// A walk that ran this round must own exactly the k1-mers on the path it recorded; if a lower rank took one of them during the
// round (before or after the walk claimed it: note_claim saw either) its record is stale: run it again.
__global__ void ext_verify_kernel(const uint8_t* __restrict__ ran, uint8_t* __restrict__ robbed, uint64_t ns, uint8_t* __restrict__ dirty) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ns) return;
  if (robbed[r]) { robbed[r] = 0; if (ran[r]) dirty[r] = 1; }
}

__global__ void ext_seed_rank_kernel(const uint32_t* __restrict__ order, uint64_t ns, Rec* __restrict__ rec) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < ns) rec[order[r]].seed_rank = (uint32_t)r;
}

// Contig bases straight from the converged claims: every oriented k1-mer knows its walk and its step index
// (claim = rank << 32 | pos; pos 0 = seed, 1..nR right steps, nR+1..nR+nL left steps), so the contig of a
// selected walk is a scatter -- no walking.  (extension_correction.py:223-245: a right step appends the last
// base of the new k1-mer, a left step prepends its first base.)
__global__ void ext_select_kernel(const uint32_t* __restrict__ ranks, uint64_t n_sel, int32_t* __restrict__ sel_of_rank,
                                  unsigned long long* __restrict__ n_twice) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_sel) return;
  if (atomicCAS((int*)&sel_of_rank[ranks[t]], -1, (int)t) != -1) atomicAdd(n_twice, 1ULL);
}

__global__ void ext_emit_claims_kernel(const u64* __restrict__ claim, uint64_t n2, const int32_t* __restrict__ sel_of_rank, uint64_t ns,
                                       const uint32_t* __restrict__ nr_a, const uint32_t* __restrict__ nl_a,
                                       const uint64_t* __restrict__ tkeys, int k, const uint64_t* __restrict__ out_off,
                                       uint8_t* __restrict__ out_bases, unsigned long long* __restrict__ counters) {
  uint32_t n_wrote = 0, n_stray = 0;
  for (uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; y < n2; y += (uint64_t)gridDim.x * blockDim.x) {
    const u64 c = claim[y];
    const uint32_t r = RANK(c), pos = POS(c);
    if (r == UNCLAIMED || r >= ns) continue;
    const int32_t t = sel_of_rank[r];
    if (t < 0) continue;
    const uint32_t nr = nr_a[r], nl = nl_a[r];
    if (nr == UNCLAIMED || pos > nr + nl) { n_stray++; continue; }    // claim of a void walk / beyond its recorded path
    uint8_t* dst = out_bases + out_off[t];
    const uint64_t str = oriented_string(tkeys, (uint32_t)y, k);
    if (pos == 0) for (int j = 0; j < k; j++) dst[nl + j] = "ACGT"[(str >> (2 * (k - 1 - j))) & 3];
    else if (pos <= nr) dst[nl + k + (pos - 1)] = "ACGT"[str & 3];
    else dst[nl - 1 - (pos - nr - 1)] = "ACGT"[(str >> (2 * (k - 1))) & 3];
    n_wrote++;
  }
  __shared__ unsigned long long blk[2];
  if (threadIdx.x < 2) blk[threadIdx.x] = 0;
  __syncthreads();
  if (n_wrote) atomicAdd(&blk[0], (unsigned long long)n_wrote);
  if (n_stray) atomicAdd(&blk[1], (unsigned long long)n_stray);
  __syncthreads();
  if (threadIdx.x < 2 && blk[threadIdx.x]) atomicAdd(&counters[threadIdx.x], blk[threadIdx.x]);
}

// ---- component shard of a k1-mer table: the walks of a connected component of the k1-mer graph touch no other
// component, so a rank that is given whole components needs only their k1-mers.  Labels the components (lock-free
// union-find over the adjacency rows), gives every component to one rank (the big ones balanced by sampled size,
// the rest by hash -- the same on every rank) and compacts this rank's k1-mers into a table of their own (same
// bucket grid, so lookups work unchanged).  Everything after that is the unsharded algorithm on the small table.
__global__ void shard_select_kernel(const uint32_t* __restrict__ lab, const uint8_t* __restrict__ owner_root, uint64_t n, uint32_t my_rank,
                                    uint32_t* __restrict__ sel) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sel[i] = owner_root[lab[i]] == my_rank ? 1u : 0u;
}
__global__ void shard_compact_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, const uint32_t* __restrict__ sel,
                                     const uint64_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ okeys, uint32_t* __restrict__ ocounts) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && sel[i]) { okeys[pos[i]] = keys[i]; ocounts[pos[i]] = counts[i]; }
}
__global__ void shard_offsets_kernel(const uint64_t* __restrict__ boff, uint64_t n_buckets, const uint64_t* __restrict__ pos,
                                     uint64_t* __restrict__ oboff) {
  uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b <= n_buckets) oboff[b] = pos[boff[b]];        // pos has n+1 entries: pos[n] = number of selected k1-mers
}

static int build_fine_dict(shn_ctx* ctx, const shn_table* t, const uint8_t* d_flags, unsigned long long** lines_out, uint64_t* n_lines_out,
                           void* room = nullptr);
static inline uint64_t fine_dict_lines(const shn_table* t) { return t->n / FD_PER_LINE + 1 + (t->layout ? t->n_buckets + 1 : 0) + FD_HOPS; }
static int component_shard(shn_ctx* ctx, const shn_table* t, int world, int rank, shn_table** out) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t n = t->n;
  uint32_t* d_weight = nullptr; uint8_t* d_flags = nullptr;
  unsigned long long* lines = nullptr;
  uint64_t n_lines = 0;
  shn_table* sub = nullptr;
  auto cleanup = [&]() { if (d_weight) shn_dev_free(d_weight); if (d_flags) shn_dev_free(d_flags); if (lines) shn_dev_free(lines); };
#define TRYS(x) do { hipError_t _e = (x); if (_e != hipSuccess) { cleanup(); if (sub) shn_table_destroy(sub); \
      return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
  TRYS(shn_dev_malloc(&d_weight, (n + 1) * 4));
  TRYS(shn_dev_malloc(&d_flags, n + 1));
  hipLaunchKernelGGL(ext_prepare_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, t->d_keys, t->d_counts, n, t->k, t->canonical, d_weight, d_flags);
  { int rca = build_fine_dict(ctx, t, d_flags, &lines, &n_lines); if (rca) { cleanup(); return rca; } }
  void *pl, *po, *pz, *pb, *pc, *pp;
  const uint32_t big_cap = 1u << 16;
  int rc;
  if ((rc = shn_ws(ctx)[14].get((n + 1) * 4, &pl)) || (rc = shn_ws(ctx)[15].get(n + 1, &po)) || (rc = shn_ws(ctx)[16].get((n + 1) * 4, &pz)) ||
      (rc = shn_ws(ctx)[17].get((size_t)big_cap * 9 + 64, &pb)) || (rc = shn_ws(ctx)[13].get(2048, &pc)) || (rc = shn_ws(ctx)[9].get((2 * n + 2) * 8, &pp))) { cleanup(); return rc; }
  uint32_t* d_lab = (uint32_t*)pl; uint8_t* d_owner_root = (uint8_t*)po;
  uint32_t* d_size = (uint32_t*)pz;
  uint32_t* d_big_root = (uint32_t*)pb; uint32_t* d_big_size = d_big_root + big_cap; uint8_t* d_big_owner = (uint8_t*)(d_big_size + big_cap);
  unsigned long long* d_cnt = (unsigned long long*)pc;
  uint64_t* d_pos = (uint64_t*)pp;
  TRYS(hipMemsetAsync(d_cnt, 0, 2048, s));
  hipLaunchKernelGGL(cc_init_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, d_lab, n);
  { TimerRegion t1(ctx, T_EXT_PREP);
    hipLaunchKernelGGL(cc_edges_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n * 8, 256), 1u << 22)), dim3(256), 0, s, shn_tab_idx(t),
                       d_flags, n, t->k, t->canonical, d_lab, (const unsigned long long*)lines, n_lines, cc_half_mode()); }
  hipLaunchKernelGGL(cc_flatten_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, d_lab, n);
  TRYS(hipMemsetAsync(d_size, 0, (n + 1) * 4, s));
  hipLaunchKernelGGL(cc_sample_kernel, dim3((uint32_t)cdiv(cdiv(n, 64), 256)), dim3(256), 0, s, d_lab, n, d_size);
  hipLaunchKernelGGL(cc_owner_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, d_lab, d_size, n, (uint32_t)world, d_owner_root,
                     d_big_root, d_big_size, d_cnt + 20, big_cap);
  unsigned long long nb = 0;
  TRYS(hipMemcpyAsync(&nb, d_cnt + 20, 8, hipMemcpyDeviceToHost, s));
  TRYS(hipStreamSynchronize(s));
  // more large components than the list holds: which ones got recorded depends on the arrival order of the atomics, i.e. could
  // differ from rank to rank -- every root then keeps its hash owner (the same on every rank), no balancing
  if (nb > big_cap) nb = 0;
  if (nb) {
    std::vector<uint32_t> br(nb), bs(nb);
    TRYS(hipMemcpyAsync(br.data(), d_big_root, nb * 4, hipMemcpyDeviceToHost, s));
    TRYS(hipMemcpyAsync(bs.data(), d_big_size, nb * 4, hipMemcpyDeviceToHost, s));
    TRYS(hipStreamSynchronize(s));
    std::vector<uint32_t> ord(nb);
    for (uint32_t j = 0; j < nb; j++) ord[j] = j;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return bs[a] != bs[b] ? bs[a] > bs[b] : br[a] < br[b]; });
    std::vector<uint64_t> load(world, 0);
    std::vector<uint8_t> bo(nb);
    for (uint32_t j : ord) {                                  // largest first onto the least loaded rank
      int best = 0;
      for (int w = 1; w < world; w++) if (load[w] < load[best]) best = w;
      bo[j] = (uint8_t)best;
      load[best] += bs[j];
    }
    TRYS(hipMemcpyAsync(d_big_owner, bo.data(), nb, hipMemcpyHostToDevice, s));       // (bo lives until the synchronisation at the end of this function)
    hipLaunchKernelGGL(cc_assign_kernel, dim3((uint32_t)cdiv(nb, 256)), dim3(256), 0, s, d_big_root, d_big_owner, (uint32_t)nb, d_owner_root);
  }
  // this rank's k1-mers, in table order (bucket by bucket, ascending inside a bucket)
  uint32_t* d_sel = d_size;                                   // (the sampled sizes are done with)
  hipLaunchKernelGGL(shard_select_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, d_lab, d_owner_root, n, (uint32_t)rank, d_sel);
  uint64_t n_sub = 0;
  if ((rc = shn_device_scan_u32(ctx, d_sel, n, d_pos, &n_sub))) { cleanup(); return rc; }
  sub = new shn_table();
  memset(sub, 0, sizeof(*sub));
  sub->ctx = t->ctx; sub->device = t->device; sub->k = t->k; sub->canonical = t->canonical; sub->n = n_sub; sub->total = 0;
  sub->bits = t->bits; sub->n_buckets = t->n_buckets; sub->layout = t->layout; sub->sk_m = t->sk_m;
  TRYS(shn_dev_malloc(&sub->d_keys, (n_sub + 1) * 8));
  TRYS(shn_dev_malloc(&sub->d_counts, (n_sub + 1) * 4));
  TRYS(shn_dev_malloc(&sub->d_bucket_off, (t->n_buckets + 1) * 8));
  hipLaunchKernelGGL(shard_compact_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, t->d_keys, t->d_counts, d_sel, d_pos, n, sub->d_keys, sub->d_counts);
  hipLaunchKernelGGL(shard_offsets_kernel, dim3((uint32_t)cdiv(t->n_buckets + 1, 256)), dim3(256), 0, s, t->d_bucket_off, t->n_buckets, d_pos, sub->d_bucket_off);
  TRYS(hipStreamSynchronize(s));
  TRYS(hipGetLastError());
#undef TRYS
  cleanup();
  *out = sub;
  return SHN_OK;
}

// the dictionary of the records / labelling kernels (see fd_build_kernel).  room: memory of at least fine_dict_lines(n) * 128 bytes
// to build it in (shn_extend: the claims and their snapshot are not in use yet -- the dictionary is 23 GB at 907 M k1-mers, and a
// block of its own on top of the records put the steady state of the K = 31 slice over the device: every step then paid for
// hipMalloc again); NULL: a block of its own, which the caller frees after the stream has drained
static int build_fine_dict(shn_ctx* ctx, const shn_table* t, const uint8_t* d_flags, unsigned long long** lines_out, uint64_t* n_lines_out,
                           void* room) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t n = t->n;
  const uint64_t n_lines = fine_dict_lines(t);
  unsigned long long* lines = (unsigned long long*)room;
  hipError_t e = lines ? hipSuccess : shn_dev_malloc(&lines, n_lines * 128);
  if (e == hipSuccess) e = hipMemsetAsync(lines, 0, n_lines * 128, s);
  if (e != hipSuccess) { if (lines && !room) shn_dev_free(lines); return shn_fail(SHN_ERR_HIP, std::string("build_fine_dict: ") + hipGetErrorString(e)); }
  static const int xcd_map = getenv("SHN_XCD_MAP") && getenv("SHN_XCD_MAP")[0] == '1';      // (off: measured below)
  if (n) hipLaunchKernelGGL(fd_build_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, shn_tab_idx(t), d_flags, n, lines, n_lines - FD_HOPS, xcd_map);
  *lines_out = lines; *n_lines_out = n_lines - FD_HOPS;          // (the look-ups hash into all but the spare lines at the end)
  return SHN_OK;
}

// ---- Component labelling on OWNER SHARDS (the N-rank path without a replicated table: BASELINE configs[4]) ----
// component_shard above wants the whole table on every rank.  Here every rank holds only the k1-mers whose minimizer it owns
// (shn_table_shard_mode 1: most edges of the k1-mer graph stay inside a shard):
//   1. shn_cc_create:   the local components -- the union-find of cc_edges_kernel over the shard (an edge whose other end is not in
//                       the shard is simply not found);
//   2. shn_cc_queries:  every neighbour / sibling key of a local k1-mer that a HIGHER rank owns, with the local root of the asker
//                       (an edge is seen from both ends; the lower rank asks, so it is recorded once) -> all-to-all by owner;
//   3. shn_cc_answer:   the owner looks the keys up: present and not low-complexity = an edge between two local components of two
//                       ranks, as a pair of global ids (rank's base + local root);
//   4. shn_cc_solve:    the edges of all ranks (gathered) -> the components of the component graph, the same on every rank: sorted
//                       distinct ids + the smallest id of the component of each;
//   5. shn_cc_labels / shn_cc_owners / shn_cc_shard: a global label and an owner rank for every local k1-mer, and the shard's pairs
//                       grouped by owner -> all-to-all -> a table of whole components per rank, walked by the unsharded shn_extend
//                       (ids and records are those of the rank's own table: the 31-bit id limit applies to a rank, not to the job).
// The edge rule is cc_edges_kernel's, so the components are those of component_shard on the whole table (tests/test_cc_shards_gpu.py
// against scipy's connected components of the same graph).
struct shn_cc {
  shn_ctx* ctx; const shn_table* t; int world, rank, device;
  uint8_t* d_flags; uint32_t* d_lab;
  unsigned long long* d_cnt;          // 64 totals + 64 bases
  uint32_t* d_bc; uint64_t* d_pos;    // per (rank, block): how many entries the block has for the rank, and where they go (see cc_query_kernel)
  uint32_t q_grid;
  uint64_t per_rank[64];
};
#define CC_GRID 2048                  // blocks of the passes that group entries by rank (count pass and write pass have the same shape)

// key number `which` = 8 * half + p of the k1-mer str (see cc_edges_kernel): half 0 = its eight neighbours, half 1 = its siblings
__device__ __forceinline__ uint64_t cc_which_key(uint64_t str, int p, int half, int k, uint64_t mask, int canonical, bool* skip) {
  const uint64_t b = (uint64_t)(p & 3);
  uint64_t key;
  *skip = false;
  if (half == 0) key = (p & 4) ? ((str >> 2) | (b << (2 * (k - 1)))) : (((str << 2) | b) & mask);
  else if (p & 4) { *skip = (str & 3) == b; key = (str & ~3ULL) | b; }
  else { const int sh = 2 * (k - 1); *skip = ((str >> sh) & 3) == b; key = (str & ~(3ULL << sh)) | (b << sh); }
  if (canonical) { const uint64_t rc = shn_revcomp(key, k); if (rc < key) key = rc; }
  return key;
}

// WRITE = false: how many queries this block has for every rank (block_count[rank * blocks + block]; the totals into `total`);
// true: the queries, grouped by rank -- pos = the exclusive scan of block_count, an LDS cursor per rank inside the block.  The two
// passes have the same launch shape.  (One HBM cursor per rank, wave-aggregated, took 292 ms for 64 M queries: every wavefront of
// the launch on the same three addresses.)
template <bool WRITE>
__global__ void cc_query_kernel(const uint64_t* __restrict__ tkeys, const uint8_t* __restrict__ flags, uint64_t n, int k, int canonical,
                                int world, int rank, const uint32_t* __restrict__ lab, unsigned long long* __restrict__ total,
                                uint32_t* __restrict__ block_count, const uint64_t* __restrict__ pos,
                                uint64_t* __restrict__ qk, uint32_t* __restrict__ ql) {
  __shared__ uint32_t lh[64];
  __shared__ uint64_t lbase[64];
  if (threadIdx.x < 64) {
    lh[threadIdx.x] = 0;
    if (WRITE) lbase[threadIdx.x] = (int)threadIdx.x < world ? pos[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] : 0;
  }
  __syncthreads();
  const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const uint64_t total_items = n * 8;
  const int m = k < SHN_OWNER_M ? k : SHN_OWNER_M, w = k - m + 1;
  const uint32_t mmask = m == 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
  for (uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total_items; gid += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = gid >> 3;
    const int p = (int)(gid & 7);
    if (flags[i] & 2) continue;                                         // (the eight lanes of a k1-mer leave together)
    const uint64_t str = tkeys[i];
    // The minimizers of the sixteen keys from the k1-mer's own m-mers (as ext_records_kernel does for the neighbours): a successor
    // and a sibling with another last base share its m-mers 1 .. w - 1 resp. 0 .. w - 2 and have one m-mer of their own, a
    // predecessor and a sibling with another first base likewise -- 14 + 16 order values per k1-mer instead of 16 x 14 (the
    // two passes of this kernel took 0.4 s of the labelling's 0.9 s at 724 M k1-mers with the minimizers made from scratch).
    uint32_t smin = 0xFFFFFFFFu, pmin = 0xFFFFFFFFu;                    // over the k1-mer's m-mers 1 .. w - 1 / 0 .. w - 2
    for (int pos = p; pos < w; pos += 8) {
      const uint32_t fm = (uint32_t)(str >> (2 * (k - m - pos))) & mmask;
      uint32_t c = fm;
      if (canonical) { const uint32_t r = shn_revcomp32(fm, m); c = r < fm ? r : fm; }
      const uint32_t o = shn_sk_order(c);
      if (pos >= 1) smin = o < smin ? o : smin;
      if (pos <= w - 2) pmin = o < pmin ? o : pmin;
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const uint32_t a = (uint32_t)__shfl_xor((int)smin, d, 64), b2 = (uint32_t)__shfl_xor((int)pmin, d, 64);
      smin = a < smin ? a : smin; pmin = b2 < pmin ? b2 : pmin;
    }
    const uint32_t nb = (uint32_t)(p & 3);
    const uint32_t first_m = (uint32_t)(str >> (2 * (k - m))) & mmask, last_m = (uint32_t)str & mmask;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      bool skip;
      const uint64_t key = cc_which_key(str, p, half, k, mask, canonical, &skip);
      if (skip) continue;
      // the key's own m-mer and which of the k1-mer's it shares
      uint32_t fm, shared;
      if (half == 0) {
        if (p & 4) { fm = (nb << (2 * (m - 1))) | (first_m >> 2); shared = pmin; }            // predecessor: new first m-mer + m-mers 0 .. w - 2
        else { fm = ((last_m & (mmask >> 2)) << 2) | nb; shared = smin; }                       // successor: m-mers 1 .. w - 1 + new last m-mer
      } else {
        if (p & 4) { fm = (last_m & ~3u) | nb; shared = pmin; }                                 // another last base: m-mers 0 .. w - 2 + its last m-mer
        else { fm = (first_m & (mmask >> 2)) | (nb << (2 * (m - 1))); shared = smin; }          // another first base: its first m-mer + m-mers 1 .. w - 1
      }
      uint32_t c = fm;
      if (canonical) { const uint32_t r = shn_revcomp32(fm, m); c = r < fm ? r : fm; }
      uint32_t o = shn_sk_order(c);
      o = shared < o ? shared : o;
      const int dest = (int)shn_owner_of_order(o, world);
      if (dest <= rank) continue;
      const uint32_t at = atomicAdd(&lh[dest], 1u);
      if (WRITE) { const uint64_t d = lbase[dest] + at; qk[d] = key; ql[d] = lab[i]; }
    }
  }
  if (!WRITE) {
    __syncthreads();
    if ((int)threadIdx.x < world) {
      block_count[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = lh[threadIdx.x];
      if (lh[threadIdx.x]) atomicAdd(&total[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
    }
  }
}

// the queries received (grouped by asking rank: group s begins at src_off[s]) against this shard
__global__ void cc_answer_kernel(const TabIdx T, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ lab,
                                 const uint64_t* __restrict__ qk, const uint32_t* __restrict__ ql, uint64_t nq,
                                 const unsigned long long* __restrict__ src_off, const unsigned long long* __restrict__ base, int world, int rank,
                                 uint64_t* __restrict__ edges, unsigned long long* __restrict__ n_edges) {
  const int lane = threadIdx.x & 63;
  const uint64_t rounded = (nq + 63) & ~63ULL;
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < rounded; q += (uint64_t)gridDim.x * blockDim.x) {
    int64_t j = -1;
    if (q < nq) { j = shn_tab_find(T, qk[q]); if (j >= 0 && (flags[j] & 2)) j = -1; }
    const unsigned long long m = __ballot(j >= 0);
    if (!m) continue;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long at = 0;
    if (lane == leader) at = atomicAdd(n_edges, (unsigned long long)__popcll(m));
    at = shfl_u64(at, leader);
    if (j >= 0) {
      int src = 0;
      while (src + 1 < world && q >= src_off[src + 1]) src++;
      const uint64_t d = at + __popcll(m & ((1ULL << lane) - 1));
      edges[2 * d] = base[rank] + lab[j];
      edges[2 * d + 1] = base[src] + ql[q];
    }
  }
}

extern "C" void shn_cc_destroy(shn_cc* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->d_flags) shn_dev_free(c->d_flags);
  if (c->d_lab) shn_dev_free(c->d_lab);
  if (c->d_cnt) shn_dev_free(c->d_cnt);
  if (c->d_bc) shn_dev_free(c->d_bc);
  if (c->d_pos) shn_dev_free(c->d_pos);
  delete c;
}

// the local components of the shard `t` of rank `rank` of `world` (t must outlive the object) + the number of queries per rank
extern "C" int shn_cc_create(shn_ctx* ctx, const shn_table* t, int world, int rank, shn_cc** out) {
  if (!ctx || !t || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return shn_fail(SHN_ERR_ARG, "shn_cc_create: bad argument");
  if (t->n >= 0x7FFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_cc_create: a shard holds at most 2^31 - 1 k1-mers (use more ranks)");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t n = t->n;
  shn_cc* c = new shn_cc();
  memset(c, 0, sizeof(*c));
  c->ctx = ctx; c->t = t; c->world = world; c->rank = rank; c->device = ctx->device;
  uint32_t* d_weight = nullptr;
  unsigned long long* lines = nullptr;
  uint64_t n_lines = 0;
  auto fail = [&](int rc) { if (d_weight) shn_dev_free(d_weight); if (lines) shn_dev_free(lines); shn_cc_destroy(c); return rc; };
#define TRYC(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e))); } while (0)
  TRYC(shn_dev_malloc(&d_weight, (n + 1) * 4));
  TRYC(shn_dev_malloc(&c->d_flags, n + 1));
  TRYC(shn_dev_malloc(&c->d_lab, (n + 1) * 4));
  TRYC(shn_dev_malloc(&c->d_cnt, 128 * 8));
  TRYC(shn_dev_malloc(&c->d_bc, (size_t)64 * CC_GRID * 4));
  TRYC(shn_dev_malloc(&c->d_pos, ((size_t)64 * CC_GRID + 2) * 8));
  TRYC(hipMemsetAsync(c->d_cnt, 0, 128 * 8, s));
  c->q_grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, cdiv(n * 8, 256)), CC_GRID);
  if (n) {
    hipLaunchKernelGGL(ext_prepare_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, t->d_keys, t->d_counts, n, t->k, t->canonical, d_weight, c->d_flags);
    { int rc = build_fine_dict(ctx, t, c->d_flags, &lines, &n_lines); if (rc) return fail(rc); }
    hipLaunchKernelGGL(cc_init_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, c->d_lab, n);
    hipLaunchKernelGGL(cc_edges_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n * 8, 256), 1u << 22)), dim3(256), 0, s, shn_tab_idx(t),
                       c->d_flags, n, t->k, t->canonical, c->d_lab, (const unsigned long long*)lines, n_lines, cc_half_mode());
    hipLaunchKernelGGL(cc_flatten_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, c->d_lab, n);
    hipLaunchKernelGGL((cc_query_kernel<false>), dim3(c->q_grid), dim3(256), 0, s, t->d_keys, c->d_flags, n, t->k, t->canonical, world, rank, c->d_lab,
                       c->d_cnt, c->d_bc, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr);
    { int rc = shn_device_scan_u32(ctx, c->d_bc, (uint64_t)world * c->q_grid, c->d_pos, nullptr); if (rc) return fail(rc); }
  }
  unsigned long long h[64];
  TRYC(hipMemcpyAsync(h, c->d_cnt, 64 * 8, hipMemcpyDeviceToHost, s));
  TRYC(hipStreamSynchronize(s));
  TRYC(hipGetLastError());
  for (int r = 0; r < 64; r++) c->per_rank[r] = r < world ? h[r] : 0;
  shn_dev_free(d_weight); d_weight = nullptr;
  if (lines) { shn_dev_free(lines); lines = nullptr; }
#undef TRYC
  *out = c;
  return SHN_OK;
}

extern "C" int shn_cc_query_counts(const shn_cc* c, uint64_t* per_rank) {
  if (!c || !per_rank) return shn_fail(SHN_ERR_ARG, "shn_cc_query_counts: bad argument");
  for (int r = 0; r < c->world; r++) per_rank[r] = c->per_rank[r];
  return SHN_OK;
}

// the queries, grouped by destination rank in rank order (per_rank[r] entries each): key (8 bytes) and the asker's local root (4 bytes)
extern "C" int shn_cc_queries(shn_cc* c, void* dev_keys_out, void* dev_labs_out) {
  if (!c) return shn_fail(SHN_ERR_ARG, "shn_cc_queries: bad argument");
  shn_ctx* ctx = c->ctx;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  unsigned long long a = 0;
  for (int r = 0; r < 64; r++) a += c->per_rank[r];
  if (!a) return SHN_OK;
  if (!dev_keys_out || !dev_labs_out) return shn_fail(SHN_ERR_ARG, "shn_cc_queries: NULL output");
  const uint64_t n = c->t->n;
  hipLaunchKernelGGL((cc_query_kernel<true>), dim3(c->q_grid), dim3(256), 0, s, c->t->d_keys, c->d_flags, n, c->t->k, c->t->canonical, c->world, c->rank,
                     c->d_lab, c->d_cnt, c->d_bc, (const uint64_t*)c->d_pos, (uint64_t*)dev_keys_out, (uint32_t*)dev_labs_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return SHN_OK;
}

// recv_per_rank[s]: queries received from rank s (grouped in rank order); base[r]: first global id of rank r (the ranks' shard sizes,
// summed); dev_edges_out: room for 2 ids per query.  n_edges: how many of the queries named a k1-mer of this shard.
extern "C" int shn_cc_answer(shn_cc* c, const void* dev_keys, const void* dev_labs, const uint64_t* recv_per_rank, const uint64_t* base,
                             void* dev_edges_out, uint64_t* n_edges) {
  if (!c || !recv_per_rank || !base || !n_edges) return shn_fail(SHN_ERR_ARG, "shn_cc_answer: bad argument");
  shn_ctx* ctx = c->ctx;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  unsigned long long h[128], a = 0;
  for (int r = 0; r < 64; r++) { h[r] = a; if (r < c->world) a += recv_per_rank[r]; h[64 + r] = r < c->world ? base[r] : 0; }
  *n_edges = 0;
  if (!a) return SHN_OK;
  if (!dev_keys || !dev_labs || !dev_edges_out) return shn_fail(SHN_ERR_ARG, "shn_cc_answer: NULL buffer");
  unsigned long long* d_ne = nullptr;
  HIP_TRY(shn_dev_malloc(&d_ne, 8));
  hipError_t e = hipMemcpyAsync(c->d_cnt, h, 128 * 8, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(d_ne, 0, 8, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cc_answer_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(a, 256), 1u << 20)), dim3(256), 0, s, shn_tab_idx(c->t), c->d_flags, c->d_lab,
                       (const uint64_t*)dev_keys, (const uint32_t*)dev_labs, (uint64_t)a, c->d_cnt, c->d_cnt + 64, c->world, c->rank,
                       (uint64_t*)dev_edges_out, d_ne);
    e = hipGetLastError();
  }
  unsigned long long ne = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&ne, d_ne, 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  shn_dev_free(d_ne);
  if (e != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_cc_answer: ") + hipGetErrorString(e));
  *n_edges = ne;
  return SHN_OK;
}

// ---- the component graph (nodes = local components that have an edge to another rank), the same computation on every rank
__global__ void ccs_iota_kernel(uint32_t* __restrict__ v, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
__global__ void ccs_first_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ first) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) first[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// pos = exclusive scan of first: the node number of sorted position i is pos[i + 1] - 1
__global__ void ccs_number_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ first,
                                  const uint64_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ node_of_end, uint64_t* __restrict__ nodes,
                                  uint32_t* __restrict__ lab) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t id = (uint32_t)(pos[i + 1] - 1);
    node_of_end[vals[i]] = id;
    if (first[i]) { nodes[id] = keys[i]; lab[id] = id; }
  }
}
__global__ void ccs_unite_kernel(const uint32_t* __restrict__ node_of_end, uint64_t n_edges, uint32_t* lab) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (uint64_t)gridDim.x * blockDim.x)
    cc_unite(lab, node_of_end[2 * e], node_of_end[2 * e + 1]);
}
__global__ void ccs_label_kernel(uint32_t* lab, const uint64_t* __restrict__ nodes, uint64_t n_nodes, uint64_t* __restrict__ label_out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += (uint64_t)gridDim.x * blockDim.x)
    label_out[i] = nodes[cc_find(lab, (uint32_t)i)];
}

// dev_edges: n_edges pairs of global ids (not modified).  dev_nodes_out / dev_labels_out: room for 2 n_edges ids each: the distinct
// ids, ascending, and for each the smallest id of its component (roots only ever link to smaller node numbers, node numbers follow
// the ids: the answer does not depend on the order of the edges).
// id_limit: every id is below it (0: unknown) -- the sort goes over its bits only.
extern "C" int shn_cc_solve(shn_ctx* ctx, const void* dev_edges, uint64_t n_edges, uint64_t id_limit, void* dev_nodes_out, void* dev_labels_out, uint64_t* n_nodes) {
  if (!ctx || !n_nodes) return shn_fail(SHN_ERR_ARG, "shn_cc_solve: bad argument");
  *n_nodes = 0;
  if (!n_edges) return SHN_OK;
  if (!dev_edges || !dev_nodes_out || !dev_labels_out) return shn_fail(SHN_ERR_ARG, "shn_cc_solve: NULL buffer");
  const uint64_t m = 2 * n_edges;
  if (m >= 0xFFFFFFF0ULL) return shn_fail(SHN_ERR_ARG, "shn_cc_solve: more than 2^31 edges between the shards");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  ShnDevBufs bufs(s);
  uint64_t *k0, *k1, *pos; uint32_t *v0, *v1, *first, *node_of_end, *lab;
  auto no = [](hipError_t e) { return e != hipSuccess; };
  if (no(bufs.get(&k0, m * 8)) || no(bufs.get(&k1, m * 8)) || no(bufs.get(&v0, m * 4)) || no(bufs.get(&v1, m * 4)) || no(bufs.get(&first, m * 4)) ||
      no(bufs.get(&pos, (m + 1) * 8)) || no(bufs.get(&node_of_end, m * 4)) || no(bufs.get(&lab, m * 4)))
    return shn_fail(SHN_ERR_HIP, "shn_cc_solve: out of device memory");
  const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(m, 256), 1u << 20);
  HIP_TRY(hipMemcpyAsync(k0, dev_edges, m * 8, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(ccs_iota_kernel, dim3(grid), dim3(256), 0, s, v0, m);
  int bit_hi = 64;
  if (id_limit) { bit_hi = 8; while (bit_hi < 64 && (id_limit >> bit_hi)) bit_hi += 8; }
  int rc = shn_sort_pairs(ctx, k0, v0, k1, v1, m, 0, bit_hi);
  if (rc) return rc;
  hipLaunchKernelGGL(ccs_first_kernel, dim3(grid), dim3(256), 0, s, k0, m, first);
  uint64_t nn = 0;
  if ((rc = shn_device_scan_u32(ctx, first, m, pos, &nn))) return rc;
  hipLaunchKernelGGL(ccs_number_kernel, dim3(grid), dim3(256), 0, s, k0, v0, first, pos, m, node_of_end, (uint64_t*)dev_nodes_out, lab);
  hipLaunchKernelGGL(ccs_unite_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n_edges, 256), 1u << 20)), dim3(256), 0, s, node_of_end, n_edges, lab);
  hipLaunchKernelGGL(ccs_label_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(nn, 256), 1u << 20)), dim3(256), 0, s, lab, (const uint64_t*)dev_nodes_out, nn,
                     (uint64_t*)dev_labels_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  *n_nodes = nn;
  return SHN_OK;
}

__device__ __forceinline__ int64_t ccs_search(const uint64_t* __restrict__ a, uint64_t n, uint64_t key) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; const uint64_t v = a[mid]; if (v == key) return (int64_t)mid; if (v < key) lo = mid + 1; else hi = mid; }
  return -1;
}
__global__ void ccs_glabel_kernel(const uint32_t* __restrict__ lab, uint64_t n, uint64_t base_me, const uint64_t* __restrict__ nodes,
                                  const uint64_t* __restrict__ labels, uint64_t n_nodes, uint64_t* __restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t g = base_me + lab[i];
    const int64_t j = ccs_search(nodes, n_nodes, g);
    out[i] = j >= 0 ? labels[j] : g;
  }
}
// the global label of every k1-mer of the shard (table order): the solved label of its local component if that has an edge to
// another rank, its own global id otherwise
extern "C" int shn_cc_labels(shn_cc* c, uint64_t base_me, const void* dev_nodes, const void* dev_labels, uint64_t n_nodes, void* dev_glabel_out) {
  if (!c) return shn_fail(SHN_ERR_ARG, "shn_cc_labels: bad argument");
  const uint64_t n = c->t->n;
  if (!n) return SHN_OK;
  if (!dev_glabel_out || (n_nodes && (!dev_nodes || !dev_labels))) return shn_fail(SHN_ERR_ARG, "shn_cc_labels: NULL buffer");
  shn_ctx* ctx = c->ctx;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  hipLaunchKernelGGL(ccs_glabel_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n, 256), 1u << 20)), dim3(256), 0, s, c->d_lab, n, base_me,
                     (const uint64_t*)dev_nodes, (const uint64_t*)dev_labels, n_nodes, (uint64_t*)dev_glabel_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return SHN_OK;
}

__global__ void ccs_owner_kernel(const uint64_t* __restrict__ glabel, uint64_t n, const uint64_t* __restrict__ big, const uint8_t* __restrict__ big_owner,
                                 uint64_t n_big, uint32_t world, uint8_t* __restrict__ owner) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t g = glabel[i];
    const int64_t j = ccs_search(big, n_big, g);
    owner[i] = j >= 0 ? big_owner[j] : (uint8_t)(shn_mix64(g ^ 0x5851F42D4C957F2DULL) % world);
  }
}
// owner rank of every k1-mer of the shard: its component's -- by the hash of the label, except for the components listed (dev_big:
// n_big labels ascending, dev_big_owner: their ranks), which the caller has balanced by size
extern "C" int shn_cc_owners(shn_cc* c, const void* dev_glabel, const void* dev_big, const void* dev_big_owner, uint64_t n_big, void* dev_owner_out) {
  if (!c) return shn_fail(SHN_ERR_ARG, "shn_cc_owners: bad argument");
  const uint64_t n = c->t->n;
  if (!n) return SHN_OK;
  if (!dev_glabel || !dev_owner_out || (n_big && (!dev_big || !dev_big_owner))) return shn_fail(SHN_ERR_ARG, "shn_cc_owners: NULL buffer");
  shn_ctx* ctx = c->ctx;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  hipLaunchKernelGGL(ccs_owner_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n, 256), 1u << 20)), dim3(256), 0, s, (const uint64_t*)dev_glabel, n,
                     (const uint64_t*)dev_big, (const uint8_t*)dev_big_owner, n_big, (uint32_t)c->world, (uint8_t*)dev_owner_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return SHN_OK;
}

template <bool WRITE>
__global__ void ccs_shard_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, const uint8_t* __restrict__ owner, uint64_t n,
                                 int world, unsigned long long* __restrict__ total, uint32_t* __restrict__ block_count, const uint64_t* __restrict__ pos,
                                 uint64_t* __restrict__ ok, uint32_t* __restrict__ oc) {
  __shared__ uint32_t lh[64];
  __shared__ uint64_t lbase[64];
  if (threadIdx.x < 64) {
    lh[threadIdx.x] = 0;
    if (WRITE) lbase[threadIdx.x] = (int)threadIdx.x < world ? pos[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] : 0;
  }
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const int o = (int)owner[i];
    const uint32_t at = atomicAdd(&lh[o], 1u);
    if (WRITE) { const uint64_t d = lbase[o] + at; ok[d] = keys[i]; oc[d] = counts[i]; }
  }
  if (!WRITE) {
    __syncthreads();
    if ((int)threadIdx.x < world) {
      block_count[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = lh[threadIdx.x];
      if (lh[threadIdx.x]) atomicAdd(&total[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
    }
  }
}
// the shard's (key, count) pairs grouped by owner rank (per_rank[r] pairs each, rank order) -- what the all-to-all sends
extern "C" int shn_cc_shard(shn_cc* c, const void* dev_owner, uint64_t* per_rank, void* dev_keys_out, void* dev_counts_out) {
  if (!c || !per_rank) return shn_fail(SHN_ERR_ARG, "shn_cc_shard: bad argument");
  const uint64_t n = c->t->n;
  for (int r = 0; r < c->world; r++) per_rank[r] = 0;
  if (!n) return SHN_OK;
  if (!dev_owner || !dev_keys_out || !dev_counts_out) return shn_fail(SHN_ERR_ARG, "shn_cc_shard: NULL buffer");
  shn_ctx* ctx = c->ctx;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, cdiv(n, 256)), CC_GRID);
  HIP_TRY(hipMemsetAsync(c->d_cnt, 0, 64 * 8, s));
  hipLaunchKernelGGL((ccs_shard_kernel<false>), dim3(grid), dim3(256), 0, s, c->t->d_keys, c->t->d_counts, (const uint8_t*)dev_owner, n, c->world, c->d_cnt,
                     c->d_bc, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr);
  int rc = shn_device_scan_u32(ctx, c->d_bc, (uint64_t)c->world * grid, c->d_pos, nullptr);
  if (rc) return rc;
  unsigned long long h[64];
  HIP_TRY(hipMemcpyAsync(h, c->d_cnt, 64 * 8, hipMemcpyDeviceToHost, s));
  hipLaunchKernelGGL((ccs_shard_kernel<true>), dim3(grid), dim3(256), 0, s, c->t->d_keys, c->t->d_counts, (const uint8_t*)dev_owner, n, c->world, c->d_cnt,
                     c->d_bc, (const uint64_t*)c->d_pos, (uint64_t*)dev_keys_out, (uint32_t*)dev_counts_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  for (int r = 0; r < c->world; r++) per_rank[r] = h[r];
  return SHN_OK;
}

extern "C" void shn_ext_destroy(shn_ext* e) {
  if (!e) return;
  hipSetDevice(e->device);
  void* ptrs[] = {e->d_weight, e->d_flags, e->d_rec, e->d_order, e->d_claim, e->d_nr, e->d_nl, e->d_totw};
  for (void* p : ptrs) if (p) shn_dev_free(p);
  if (e->owned_table) shn_table_destroy(e->owned_table);
  delete e;
}

// Pipelining hook: called from inside shn_extend, on the calling thread, whenever the walks of a rank block are final
// (lo <= rank < hi; their claims never change again), so that the caller can take their contigs and start the contig stage
// while the later blocks are still iterating.  status 0 = a block, 1 = the last block, -1 = the fixpoint audit reopened
// the blocks: everything handed over so far is void.  The shn_ext passed is valid for shn_ext_stats_range / shn_ext_emit /
// shn_ext_seed_info during the call-back.
typedef void (*shn_block_cb)(void* user, shn_ext* e, uint64_t lo, uint64_t hi, int status);
static thread_local shn_block_cb g_block_cb = nullptr;
static thread_local void* g_block_user = nullptr;
extern "C" void shn_ext_set_block_callback(shn_block_cb cb, void* user) { g_block_cb = cb; g_block_user = user; }

// checksum of one array into e->dig[stage] (added to what is there: a stage may be made of several arrays)
static int ext_digest(shn_ctx* ctx, shn_ext* e, int stage, const void* d, uint64_t bytes, uint64_t salt) {
  if (!bytes) return SHN_OK;
  hipStream_t s = ctx->stream;
  unsigned long long* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, EXT_DIG_CHUNKS * 8));
  hipError_t er = hipMemsetAsync(d_out, 0, EXT_DIG_CHUNKS * 8, s);
  uint64_t h[EXT_DIG_CHUNKS];
  if (er == hipSuccess) {
    hipLaunchKernelGGL(ext_digest_kernel, dim3(1024), dim3(256), 0, s, (const uint32_t*)d, bytes / 4, salt, d_out);
    er = hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, s);
  }
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  (void)hipFree(d_out);
  if (er != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("ext_digest: ") + hipGetErrorString(er));
  for (int i = 0; i < EXT_DIG_CHUNKS; i++) e->dig[stage][i] += h[i];
  e->has_dig = 1;
  return SHN_OK;
}
extern "C" int shn_ext_digests(const shn_ext* e, uint64_t* out) {
  if (!e || !out) return shn_fail(SHN_ERR_ARG, "shn_ext_digests: NULL argument");
  if (!e->has_dig) return shn_fail(SHN_ERR_ARG, "shn_ext_digests: the extension was not made with SHN_EXT_DIGEST=1");
  memcpy(out, e->dig, sizeof e->dig);
  return SHN_OK;
}

extern "C" int shn_extend(shn_ctx* ctx, const shn_table* t, uint32_t min_weight, int max_iterations, shn_ext** out) {
  return shn_extend_sharded(ctx, t, min_weight, max_iterations, 1, 0, out);
}

extern "C" int shn_extend_sharded(shn_ctx* ctx, const shn_table* t, uint32_t min_weight, int max_iterations, int world, int rank,
                                  shn_ext** out) {
  if (!ctx || !t || !out) return shn_fail(SHN_ERR_ARG, "shn_extend: NULL argument");
  if (world < 1 || world > 255 || rank < 0 || rank >= world) return shn_fail(SHN_ERR_ARG, "shn_extend_sharded: bad world/rank");
  if (2 * t->n >= 0x7FFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_extend: table too large for 31-bit oriented ids");
  SHN_ENTER(ctx);
  shn_stage_begin(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  if (world > 1 && t->n) {
    shn_table* sub = nullptr;
    int rcs;
    { TimerRegion treg(ctx, T_EXTEND); rcs = component_shard(ctx, t, world, rank, &sub); }
    if (rcs) return rcs;
    rcs = shn_extend_sharded(ctx, sub, min_weight, max_iterations, 1, 0, out);
    if (rcs) { shn_table_destroy(sub); return rcs; }
    (*out)->owned_table = sub;
    return SHN_OK;
  }
  TimerRegion treg(ctx, T_EXTEND);
  shn_ext* e = new shn_ext();
  memset(e, 0, sizeof(*e));
  e->ctx = ctx; e->device = ctx->device; e->k = t->k; e->n = t->n; e->min_weight = min_weight; e->table = t;
  uint64_t n = t->n;
  if (max_iterations <= 0) max_iterations = 100000;
#define TRYE(x) do { hipError_t _e = (x); if (_e != hipSuccess) { shn_ext_destroy(e); \
      return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
  TRYE(shn_dev_malloc(&e->d_weight, (n + 1) * 4));
  TRYE(shn_dev_malloc(&e->d_flags, n + 1));
  TRYE(shn_dev_malloc(&e->d_rec, (2 * n + 1) * sizeof(Rec)));
  // claims and snapshot in one block ((+4: the begin pass reads four claims per thread); the dictionary of the records kernel is
  // built in it first, see build_fine_dict)
  const uint64_t claim_words = (2 * n + 4 + 31) & ~31ULL;
  { u64* both = nullptr;
    TRYE(shn_dev_malloc(&both, std::max<uint64_t>(2 * claim_words * 8, fine_dict_lines(t) * 128)));
    e->d_claim = both; e->d_claim2 = both + claim_words; }
  if (n) {
    TimerRegion t1(ctx, T_EXT_PREP);
    hipLaunchKernelGGL(ext_prepare_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, t->d_keys, t->d_counts, n, t->k,
                       t->canonical, e->d_weight, e->d_flags);
    {
      unsigned long long* lines = nullptr;
      uint64_t n_lines = 0;
      { int rca = build_fine_dict(ctx, t, e->d_flags, &lines, &n_lines, e->d_claim); if (rca) { shn_ext_destroy(e); return rca; } }
      // (off by default -- MEASURED at BASELINE configs[2], one box: 144 ms with two dictionary look-ups in flight, 170 ms with four
      // (one wavefront per SIMD fewer), against 119.5 ms for the dictionary-only kernel, same records.  The look-ups that are left
      // come one or two at a time behind their bucket offsets where the kernel above has eight in flight, and a trip's chain of
      // bisection, ballots and shuffles is longer than the one fetch it replaces: fewer bytes, more latency.  HISTORY.md, round 5.)
      static const bool rec_lds = getenv("SHN_REC_LDS") && getenv("SHN_REC_LDS")[0] == '1';
      if (t->layout == 1 && rec_lds && t->n_buckets / REC_G + 1 < 0x7FFFFFFFULL) {
        TimerRegion ta(ctx, T_EXT_ADJ);
        hipLaunchKernelGGL(ext_records_lds_kernel, dim3((uint32_t)cdiv(t->n_buckets, REC_G)), dim3(256), 0, s, shn_tab_idx(t), e->d_flags, e->d_weight,
                           (uint64_t)t->n_buckets, t->k, t->canonical, e->d_rec, (const unsigned long long*)lines,
                           getenv("SHN_REC_ABLATE") ? atoi(getenv("SHN_REC_ABLATE")) : 0);
      } else
      { TimerRegion ta(ctx, T_EXT_ADJ);                  // (one launch: bench.py's roofline entry for this kernel)
        hipLaunchKernelGGL(ext_records_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(n * 8, 256), 1u << 22)), dim3(256), 0, s, shn_tab_idx(t),
                           e->d_flags, e->d_weight, n, t->k, t->canonical, e->d_rec, (const unsigned long long*)lines, n_lines,
                           (getenv("SHN_XCD_MAP") && getenv("SHN_XCD_MAP")[0] == '1') ? 1 : 0); }
      (void)lines;                                       // (lives in the claims' block: overwritten when the claims are initialised below)
    }
    TRYE(hipGetLastError());
  }
  // seeds: compact, sort by string then (stable) by weight descending
  void *pk, *pv, *pk2, *pv2, *pc;
  int rc;
  if ((rc = shn_ws(ctx)[13].get(2048, &pc))) { shn_ext_destroy(e); return rc; }
  unsigned long long* d_cnt = (unsigned long long*)pc;      // [0] seeds [1] steps [2..4] plan (long, pool, short) [5] final pool [6] changed
  // the seeds are counted first: the sort buffers are sized for them, not for every oriented k1-mer (at 20,000 genes 13 % of
  // the table are seeds -- 30 GB less)
  TRYE(hipMemsetAsync(d_cnt, 0, 2048, s));
  const uint64_t n_sblk = cdiv(n, 1024);
  uint32_t* d_bcnt = nullptr; uint64_t* d_bbase = nullptr;
  struct SeedScratch { uint32_t** a; uint64_t** b; ~SeedScratch() { if (*a) shn_dev_free(*a); if (*b) shn_dev_free(*b); } } seed_scratch{&d_bcnt, &d_bbase};
  TRYE(shn_dev_malloc(&d_bcnt, (n_sblk + 1) * 4));
  TRYE(shn_dev_malloc(&d_bbase, (n_sblk + 2) * 8));
  if (n) hipLaunchKernelGGL(ext_seed_kernel, dim3((uint32_t)n_sblk), dim3(1024), 0, s, t->d_keys, e->d_weight, e->d_flags, n,
                            t->k, t->canonical, min_weight, d_bcnt, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr);
  uint64_t ns = 0;
  if (n && (rc = shn_device_scan_u32(ctx, d_bcnt, n_sblk, d_bbase, &ns))) { shn_ext_destroy(e); return rc; }
  if ((rc = shn_ws(ctx)[9].get((ns + 2) * 8, &pk)) || (rc = shn_ws(ctx)[10].get((ns + 2) * 4, &pv)) ||
      (rc = shn_ws(ctx)[11].get((ns + 2) * 8, &pk2)) || (rc = shn_ws(ctx)[12].get((ns + 2) * 4, &pv2))) { shn_ext_destroy(e); return rc; }
  uint64_t* skeys = (uint64_t*)pk; uint32_t* svals = (uint32_t*)pv;
  if (n) hipLaunchKernelGGL(ext_seed_kernel, dim3((uint32_t)n_sblk), dim3(1024), 0, s, t->d_keys, e->d_weight, e->d_flags, n,
                            t->k, t->canonical, min_weight, d_bcnt, (const uint64_t*)d_bbase, skeys, svals);
  e->n_seeds = ns;
  {
    TimerRegion t2(ctx, T_EXT_SORT);
    if ((rc = shn_sort_pairs(ctx, skeys, svals, (uint64_t*)pk2, (uint32_t*)pv2, ns, 0, 2 * t->k))) { shn_ext_destroy(e); return rc; }
    if (ns) {
      hipLaunchKernelGGL(ext_weightkey_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, svals, e->d_weight, ns, skeys);
      if ((rc = shn_sort_pairs(ctx, skeys, svals, (uint64_t*)pk2, (uint32_t*)pv2, ns, 0, 32))) { shn_ext_destroy(e); return rc; }
    }
  }
  TRYE(shn_dev_malloc(&e->d_order, (ns + 1) * 4));
  TRYE(shn_dev_malloc(&e->d_nr, (ns + 1) * 4));
  TRYE(shn_dev_malloc(&e->d_nl, (ns + 1) * 4));
  TRYE(shn_dev_malloc(&e->d_totw, (ns + 1) * 8));
  // memo slots are never recycled within a call (a word per step ever walked by a memo-bearing walk); pool
  // indices live in 30 bits of a hint
  uint64_t pool_cap = std::min<uint64_t>(24 * n + (1ULL << 20), (1ULL << 30) - 1);
  if (getenv("SHN_EXT_POOL_WORDS")) pool_cap = std::max<uint64_t>(256, std::min<uint64_t>(pool_cap, strtoull(getenv("SHN_EXT_POOL_WORDS"), nullptr, 10)));   // (tests: a full pool only costs time)
  TRYE(hipMemcpyAsync(e->d_order, svals, ns * 4, hipMemcpyDeviceToDevice, s));
  TRYE(hipMemsetAsync(e->d_nr, 0xFF, (ns + 1) * 4, s));
  TRYE(hipMemsetAsync(e->d_nl, 0, (ns + 1) * 4, s));
  TRYE(hipMemsetAsync(e->d_totw, 0, (ns + 1) * 8, s));
  TRYE(hipMemsetAsync(e->d_claim, 0xFF, (2 * n + 4) * 8, s));
  // scratch: claim snapshot (d_claim2), memo pool + per-k1-mer hints, per-walk plan arrays
  void *ppool, *pplan;
  if ((rc = shn_ws(ctx)[27].get(pool_cap * 4, &ppool)) ||
      (rc = shn_ws(ctx)[28].get((ns + 1) * (8 + 4 * 10 + 1 + 1 + 1 + 1 + 1 + 1 + 1) + 64, &pplan))) { shn_ext_destroy(e); return rc; }
  u64 *claim = e->d_claim, *snap = e->d_claim2;
  uint32_t* pool = (uint32_t*)ppool;
  uint64_t* moff = (uint64_t*)pplan;
  uint32_t* mcap = (uint32_t*)(moff + ns + 1);
  uint32_t* mR = mcap + ns + 1;
  uint32_t* mL = mR + ns + 1;
  uint32_t* long_list = mL + ns + 1;
  uint32_t* short_list = long_list + ns + 1;
  uint32_t* promo_list = short_list + ns + 1;
  uint32_t* res_cur = promo_list + ns + 1;
  uint32_t* res_info = res_cur + ns + 1;
  uint32_t* owned = res_info + ns + 1;
  uint32_t* log_head = owned + ns + 1;                       // claim logs: a walk's last chunk (LOG_LOST: none)
  uint8_t* mvalid = (uint8_t*)(log_head + ns + 1);
  uint8_t* fill = mvalid + ns + 1;
  uint8_t* dirty = fill + ns + 1;
  uint8_t* ran = dirty + ns + 1;
  uint8_t* robbed = ran + ns + 1;
  uint8_t* settled = robbed + ns + 1;     // walks that can never survive (ext_chain_has_lower): void for good, never launched
  uint8_t* log_cnt = settled + ns + 1;    // claim logs: entries of a walk's last chunk
  TRYE(hipMemsetAsync(robbed, 0, 2 * (ns + 1), s));
  TRYE(hipMemsetAsync(log_head, 0xFE, (ns + 1) * 4, s));
  uint8_t* robsat = nullptr;              // (development, SHN_EXT_XTIME: walks robbed while they sat out, until they run again)
  struct RobsatFree { uint8_t** p; ~RobsatFree() { if (*p) shn_dev_free(*p); } } robsat_free{&robsat};
  if (getenv("SHN_EXT_XTIME")) { TRYE(shn_dev_malloc(&robsat, ns + 1)); TRYE(hipMemsetAsync(robsat, 0, ns + 1, s)); }
  TRYE(hipMemsetAsync(mvalid, 0, 2 * (ns + 1), s));
  TRYE(hipMemsetAsync(pool, 0xFF, pool_cap * 4, s));            // NONE32: "no entry"
  // (hints and seed ranks live in the records: ext_records_kernel wrote "none" into both)
  if (ns) hipLaunchKernelGGL(ext_seed_rank_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, e->d_order, (uint64_t)ns, e->d_rec);
  const bool want_dig = getenv("SHN_EXT_DIGEST") && getenv("SHN_EXT_DIGEST")[0] == '1';
  if (want_dig) {
    int rd;
    if ((rd = ext_digest(ctx, e, 0, t->d_keys, n * 8, 1)) || (rd = ext_digest(ctx, e, 1, t->d_counts, n * 4, 2)) ||
        (rd = ext_digest(ctx, e, 2, t->d_bucket_off, (t->n_buckets + 1) * 8, 3)) || (rd = ext_digest(ctx, e, 3, e->d_weight, n * 4, 4)) ||
        (rd = ext_digest(ctx, e, 3, e->d_flags, n & ~3ULL, 5)) || (rd = ext_digest(ctx, e, 4, e->d_rec, 2 * n * sizeof(Rec), 6)) ||
        (rd = ext_digest(ctx, e, 5, e->d_order, ns * 4, 7))) { shn_ext_destroy(e); return rd; }
  }
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  TRYE(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  TRYE(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  TRYE(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  static thread_local unsigned long long* plan = nullptr;      // long dirty walks, -, short dirty walks, dirty walks
  if (!plan) TRYE(hipHostMalloc((void**)&plan, 64));
  int it = 0, repairs = 0;
  bool converged = ns == 0;
  uint8_t* chunk = nullptr;               // see ext_mark_kernel
  const uint64_t n_chunks = ((2 * n) >> CHUNK_SHIFT) + 16;
  TRYE(shn_dev_malloc(&chunk, n_chunks));
  struct ChunkFree { uint8_t* p; ~ChunkFree() { shn_dev_free(p); } } chunk_free{chunk};
  TRYE(hipMemsetAsync(chunk, 0, n_chunks, s));
  uint8_t* coarse = nullptr;              // see ext_plan_kernel
  TRYE(shn_dev_malloc(&coarse, (size_t)ns / 64 + 64));
  struct CoarseFree { uint8_t* p; ~CoarseFree() { shn_dev_free(p); } } coarse_free{coarse};
  bool snap_current = false;              // the snapshot equals the claims (after a round with precise marks)
  bool fresh_block = true;                // the open block has not run a round yet (and no repair has reopened earlier blocks)

  const uint32_t g2n = (uint32_t)cdiv(2 * n, 256);
  // Rank phases: a walk depends only on lower ranks, so the fixpoint is reached block by block -- first the
  // heaviest seeds (where the long, mutually dependent walks live), then geometrically larger blocks that see
  // final lower ranks and settle in a few rounds.
  // (many seeds -- BASELINE configs[2]: 186 M over 20,000 genes -- interfere locally: fewer, larger blocks; every round costs two
  // passes over all claims whatever it re-runs.  tools/ext_blocks_probe.py: 51 rounds / 2.27 s -> 32 rounds / 2.15 s)
  const bool many = ns >= (1ULL << 24);
  unsigned long long lim0 = std::max<unsigned long long>(ns / (many ? 8 : 32), 4096), grow = 4;
  if (getenv("SHN_EXT_LIMIT0")) lim0 = strtoull(getenv("SHN_EXT_LIMIT0"), nullptr, 10);
  if (getenv("SHN_EXT_GROW")) grow = strtoull(getenv("SHN_EXT_GROW"), nullptr, 10);
  const unsigned long long tail_div = getenv("SHN_EXT_TAIL") ? strtoull(getenv("SHN_EXT_TAIL"), nullptr, 10) : (many ? 0 : 8);   // last blocks = ns / 8 walks (0: off)
  uint32_t frozen = 0, limit = (uint32_t)std::min<unsigned long long>(ns, lim0);
  TRYE(hipMemsetAsync(dirty, 0, 2 * (ns + 1), s));               // dirty + ran
  TRYE(hipMemsetAsync(dirty, 1, limit, s));
  TRYE(hipMemsetAsync(owned, 0, (ns + 1) * 4, s));
  plan[6] = 64;                                                  // memo pool cursor: 64 words of NONE32 padding in front
  TRYE(hipMemcpyAsync(d_cnt + 10, plan + 6, 8, hipMemcpyHostToDevice, s));   // (from the pinned block, on the context's stream like everything else here)
  auto tune = [](const char* name, uint32_t dflt) { const char* v = getenv(name); return v ? (uint32_t)strtoul(v, nullptr, 10) : dflt; };
  const int precise_marks = (int)tune("SHN_EXT_PRECISE", 1);       // 0: the conservative rule (every walk standing next to a freed k1-mer)
  const uint32_t long_walk = tune("SHN_EXT_LONG_WALK", LONG_WALK), memo_min = tune("SHN_EXT_MEMO_MIN", MEMO_MIN),
                 promote_steps = tune("SHN_EXT_PROMOTE", PROMOTE_STEPS);
  // Bulk rounds: with hundreds of thousands of dirty walks the GPU is throughput-bound, not latency-bound, and one thread
  // per walk (one memory round trip per step, every lane busy) beats a wavefront per walk by an order of magnitude; memos
  // (which serve the latency-bound re-runs of a few long walks) are not made in such a round.  The expected number of
  // dirty walks is the block size when a block opens, else the count of the round before.
  const unsigned long long bulk_min = getenv("SHN_EXT_BULK") ? strtoull(getenv("SHN_EXT_BULK"), nullptr, 10) : 262144ULL;
  const unsigned long long dense_min = getenv("SHN_EXT_DENSE") ? strtoull(getenv("SHN_EXT_DENSE"), nullptr, 10) : (4ULL << 20);   // (BASELINE configs[2]: 262144 -> 954 ms, 2 M or 16 M -> 900 ms per extension)
  const int seed_check = (int)tune("SHN_EXT_SEEDCHECK", 1);
  const int first_look = (int)tune("SHN_EXT_FIRST_LOOK", 0);      // (measured at configs[2]: first-round launches 172-174 ms with it, 163-170 without: off)
  // bulk rounds: a thread walker that gets this far hands its walk to the packed second launch (0: it walks to the end itself, as until round 4)
  const uint32_t bulk_promote = tune("SHN_EXT_PROMOTE_BULK", 0);      // (measured at BASELINE configs[2], round 5: 8 / 24 / 64 -> walk kernels 273 / 267 / 268 ms per step against 224 without -- the bulk rounds are bound by the random fetches of their steps, not by idle lanes; off)
  const unsigned long long resume_waves = tune("SHN_EXT_RESUME_WAVES", 8192);
  const bool skip_idle_begin = tune("SHN_EXT_MEMO_RELEASE", 1) != 0;          // rounds whose dirty walks all have a current memo release through it (ext_release_memo_kernel)
  const unsigned long long memo_release_max = tune("SHN_EXT_MEMO_RELEASE_MAX", 65536);
  int n_begin_skipped = 0;
  const bool prepass = tune("SHN_EXT_PREPASS", 1) != 0;
  // forced links followed on each side of a seed when its block opens (ext_chain_has_lower).  MEASURED at BASELINE configs[2] (round 6, one
  // box, 4 steps each): hops 0 / 1 / 2 / 4 / 8 / 16 / 64 -> 0 / 78 / 88 / 96 / 99 / 100 / 100.5 M of the 186 M walks never launched, same
  // digest, first-round walker 164 / 208 / 193 / 163 / 158 / 159 / 152 ms, walk steps 627 / 957 / 891 / 738 / 702 / 696 / 664 M, the walks'
  // host time 0.608 / 0.687 / 0.671 / 0.638 / 0.629 / 0.632 / 0.631 s: the rule is exact and removes half of the walks -- the cheap half
  // (seeds on clean chains: weakly covered transcripts; a well covered one has an error branch at every position, so no link is
  // forced) -- and the survivors, started earlier than they would have been behind the void ones, walk further before a lower rank
  // stops them.  A launch's time follows its steps' random lines (~5.5 per step at ~20 G lines/s; 623 M steps for the 422 M k1-mers finally claimed:
  // tools/walk_waste_r06.py), not its walks.  Off by default.
  const uint32_t settle_hops = tune("SHN_EXT_SETTLE_HOPS", 0);
  // Claim logs (round 6; SHN_EXT_LOGS=0: off).  The begin pass streams every claim (11.6 GB at BASELINE configs[2], 2.7 ms) to find
  // those of the walks that re-run; 22 of a step's 25 rounds re-run walks that hold fewer than ten million claims between them, and
  // what kept them on the stream was a handful of claim holders per round WITHOUT a memo: walks that last ran in a bulk round (no
  // memos there: rebuilding them from the claims is a scattered store per claim).  Their chains are intact but thousands of steps
  // long -- following one from its seed is two dependent round trips a step (SHN_EXT_XTIME=1 says how many and how long).  So the
  // bulk walker writes the k1-mers it claims into a log of its own as it goes (a 4-byte store per step into a 32-byte chunk; a
  // wavefront reserves 64 chunks with one atomic), and a round whose claim holders all have a current memo or a log, and few
  // enough claims to give back, releases through those (ext_release_memo_kernel) instead of the stream.
  const bool use_logs = tune("SHN_EXT_LOGS", 1) != 0 && bulk_promote == 0 && ns > 0;
  const unsigned long long targeted_max_steps = getenv("SHN_EXT_TARGETED_MAX") ? strtoull(getenv("SHN_EXT_TARGETED_MAX"), nullptr, 10) : (12ULL << 20);
  uint32_t* logpool = nullptr;
  struct LogFree { uint32_t** p; ~LogFree() { if (*p) shn_dev_free(*p); } } log_free{&logpool};
  uint64_t log_cap = use_logs ? std::min<uint64_t>(2 * n / 4 + (1u << 16), 0x7FFFFFF0ULL) : 0;
  if (use_logs && getenv("SHN_EXT_LOG_CHUNKS")) log_cap = std::max<uint64_t>(LOG_SLAB, std::min<uint64_t>(log_cap, strtoull(getenv("SHN_EXT_LOG_CHUNKS"), nullptr, 10)));   // (tests: a pool that runs out -- the walks' logs are void and their rounds fall back to the begin pass)
  if (use_logs) { TRYE(shn_dev_malloc(&logpool, log_cap * LOG_WORDS * 4)); TRYE(hipMemsetAsync(d_cnt + 26, 0, 8, s)); }
  const uint32_t fresh_split = std::max<uint32_t>(1, std::min<uint32_t>(16, tune("SHN_EXT_FRESH_SPLIT", 1)));   // sub-launches of a block's first (bulk) round (measured at configs[2]: 1 / 4 / 7 / 10 -> 184 / 176 / 209 / 248 ms: every sub-launch waits for its longest walk; off)
  const uint32_t fresh_split_min = tune("SHN_EXT_FRESH_SPLIT_MIN", 65536);                                       // ... of blocks of at least this many walks            // a block's first round settles the walks whose seed an earlier block holds (ext_plan_kernel)
  unsigned long long expect_dirty = limit;
  while (!converged && it < max_iterations) {
    const bool bulk = bulk_min && expect_dirty >= bulk_min;
    // dense: the begin / mark passes stream all claims (rounds that write nearly everywhere); else they follow the 128-byte-line
    // flags the walkers and the release leave behind.  A bulk round of a few hundred thousand walks writes a few million
    // k1-mers: far fewer lines than the 23 GB of claims and snapshot the dense passes read.
    const bool dense = bulk && expect_dirty >= dense_min;
    // classify the dirty walks of the open block; a block without dirty walks is consistent = final
    TRYE(hipMemsetAsync(d_cnt + 2, 0, 32, s));
    TRYE(hipMemsetAsync(d_cnt + 24, 0, 8, s));
    if (limit > frozen)
      hipLaunchKernelGGL(ext_plan_kernel, dim3((uint32_t)cdiv(limit - frozen, 1024)), dim3(1024), 0, s, e->d_nr, e->d_nl, (uint64_t)limit, frozen,
                         mvalid, mR, mL, dirty, long_list, short_list, d_cnt + 2, bulk ? 0xFFFFFFFFu : long_walk, coarse,
                         (fresh_block && frozen > 0 && prepass) ? (const u64*)claim : (const u64*)nullptr, e->d_order, e->d_totw,
                         (const Rec*)e->d_rec, fresh_block ? settle_hops : 0u, settled, d_cnt + 20, robsat, d_cnt + 21,
                         use_logs ? (const uint32_t*)log_head : (const uint32_t*)nullptr, d_cnt + 24);
    // (pinned host memory: a pageable destination costs a staging copy kernel per round)
    TRYE(hipMemcpyAsync(plan, d_cnt + 2, 32, hipMemcpyDeviceToHost, s));
    TRYE(hipMemcpyAsync(plan + 7, d_cnt + 24, 8, hipMemcpyDeviceToHost, s));      // the claims the dirty walks hold (by their records)
    TRYE(hipStreamSynchronize(s));
    expect_dirty = plan[3];
    if (plan[3] == 0) {
      if (limit >= ns) {
        // every block is settled: audit the claims (see ext_audit_nodes_kernel); a walk that is not at its fixpoint
        // re-runs with all blocks open.  Never seen to fire in testing except by fault injection (SHN_EXT_FAULT).
        TimerRegion ta(ctx, T_EXT_AUDIT);
        TRYE(hipMemsetAsync(owned, 0, (ns + 1) * 4, s));
        TRYE(hipMemsetAsync(d_cnt + 48, 0, 16, s));
        hipLaunchKernelGGL(ext_audit_nodes_kernel, dim3(std::min<uint32_t>(g2n, 4096u)), dim3(256), 0, s, claim, 2 * n, rows_R(e->d_rec),
                           rows_L(e->d_rec), words_weight(e->d_rec), e->d_order, e->d_nr, e->d_nl, (uint64_t)ns, owned, dirty, d_cnt + 48);
        hipLaunchKernelGGL(ext_audit_walks_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, claim, e->d_order, e->d_nr, e->d_nl, (uint64_t)ns,
                           owned, dirty, d_cnt + 48);
        TRYE(hipMemcpyAsync(plan + 4, d_cnt + 48, 16, hipMemcpyDeviceToHost, s));
        TRYE(hipStreamSynchronize(s));
        if (plan[4] == 0 && plan[5] == 0) {
          converged = true;
          if (g_block_cb) g_block_cb(g_block_user, e, frozen, ns, 1);
          break;
        }
        if (g_block_cb) g_block_cb(g_block_user, e, 0, 0, -1);
        fprintf(stderr, "[shn_extend] fixpoint audit after %d rounds: %llu k1-mers / %llu walks disagree with the greedy rule; reopening all blocks\n",
                it, plan[4], plan[5]);
        if (++repairs > 16) break;
        fresh_block = false;
        frozen = 0;
        expect_dirty = plan[4] + plan[5];
        TRYE(hipMemsetAsync(ran, 0, ns + 1, s));
        TRYE(hipMemsetAsync(owned, 0, (ns + 1) * 4, s));
        continue;
      }
      if (g_block_cb && repairs == 0) g_block_cb(g_block_user, e, frozen, limit, 0);
      frozen = limit;                                // this block is final: open the next one
      // geometric blocks up to half of the walks, then blocks of ns / tail_div: the light half of the seed order settles in
      // a few cheap rounds per block, and what a block call-back receives early can be worked on beside the later blocks
      if (tail_div && ((unsigned long long)frozen + ns / 16) * 2 >= ns) limit = (uint32_t)std::min<unsigned long long>(ns, (unsigned long long)frozen + std::max<unsigned long long>(1, ns / tail_div));
      else limit = (uint32_t)std::min<unsigned long long>(ns, (unsigned long long)limit * grow);
      TRYE(hipMemsetAsync(ran, 0, frozen, s));       // frozen walks never run again
      TRYE(hipMemsetAsync(dirty + frozen, 1, limit - frozen, s));
      expect_dirty = limit - frozen;
      fresh_block = repairs == 0;
      continue;
    }
    TimerRegion t3(ctx, T_EXT_WALK);
    // snapshot, then release the claims of the walks that re-run this round
    // (a block that has just opened holds no claims yet: with the snapshot up to date there is nothing to release and nothing to copy)
    // (... and so does a round none of whose dirty walks holds a claim -- void walks looking at their seed again: plan[1])
    const bool memo_release = !fresh_block && !dense && skip_idle_begin && plan[1] == 0 &&
                              (use_logs ? plan[7] <= targeted_max_steps : plan[0] + plan[2] <= memo_release_max);
    const bool snap_was_current = snap_current;
    const bool memo_release_done = !fresh_block && memo_release && precise_marks && snap_current;
    if ((fresh_block || memo_release) && precise_marks && snap_current) {
      TRYE(hipMemsetAsync(d_cnt + 6, 0, 16, s)); TRYE(hipMemsetAsync(d_cnt + 13, 0, 16, s));
      if (!fresh_block) {
        n_begin_skipped++;
        if (plan[0] + plan[2])
          hipLaunchKernelGGL(ext_release_memo_kernel, dim3((uint32_t)(plan[0] + plan[2])), dim3(64), 0, s, long_list, (uint64_t)plan[0], short_list, (uint64_t)plan[2],
                             e->d_nr, moff, mR, mL, pool, claim, chunk, (const uint8_t*)mvalid, (const uint32_t*)e->d_order, (const uint32_t*)logpool,
                             (const uint32_t*)log_head, (const uint8_t*)log_cnt);
      }
    }
    else {
      TimerRegion tb(ctx, T_EXT_BEGIN);
      hipLaunchKernelGGL(ext_round_begin_kernel, dim3((uint32_t)cdiv(cdiv(2 * n, 4), 256)), dim3(256), 0, s, claim, snap, 2 * n, dirty, (uint64_t)ns, d_cnt,
                         (!precise_marks || !snap_current) ? 1 : 0, dense ? (uint8_t*)nullptr : chunk, frozen, limit, coarse);
    }
    snap_current = true;
    WalkArgs A;
    A.order = e->d_order; A.adjR = rows_R(e->d_rec); A.adjL = rows_L(e->d_rec); A.weight = words_weight(e->d_rec);
    A.claim = claim; A.claim_old = snap;
    A.nr_out = e->d_nr; A.nl_out = e->d_nl; A.totw_out = e->d_totw;
    A.pool = pool; A.moff = moff; A.mR = mR; A.mL = mL; A.mvalid = mvalid; A.hint = words_hint(e->d_rec);
    A.promote_steps = bulk ? (bulk_promote ? bulk_promote : 0xFFFFFFFFu) : promote_steps;
    A.promo_list = promo_list; A.promo_count = d_cnt + 13; A.res_cur = res_cur; A.res_info = res_info;
    A.chunk = dense ? nullptr : chunk;         // (dense rounds write nearly everywhere: their mark pass is dense, the walkers do not flag)
    A.robbed = robbed;
    A.seed_check = seed_check;
    A.first_look = first_look;
    A.logpool = (use_logs && bulk) ? logpool : nullptr; A.log_head = log_head; A.log_cnt = log_cnt; A.log_cursor = d_cnt + 26; A.log_cap = log_cap;
    A.steps_counter = d_cnt + 1; A.fresh_steps_counter = d_cnt + 16; A.wave_steps_counter = d_cnt + 64; A.dbg = (getenv("SHN_DEBUG") || getenv("SHN_EXT_XTIME")) ? d_cnt + 32 : nullptr;
    // long (wave per walk) and short (thread per walk) kernels are independent: overlap them on two streams
    if (plan[0]) {
      TRYE(hipEventRecord(ev_fork, s));
      TRYE(hipStreamWaitEvent(aux, ev_fork, 0));
      { TimerRegion tk(ctx, T_EXT_WALK_WAVE, aux);
        hipLaunchKernelGGL(ext_walk_long_kernel<false>, dim3((uint32_t)plan[0]), dim3(64), 0, aux, A, long_list, (uint64_t)ns, (const unsigned long long*)nullptr); }
      TRYE(hipEventRecord(ev_join, aux));
    }
    double x_t0 = 0;
    if (getenv("SHN_EXT_XTIME")) {   // (development: time of every thread-walker launch)
       TRYE(hipStreamSynchronize(s)); timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); x_t0 = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    const bool was_fresh = fresh_block;
    if (fresh_block && bulk && fresh_split > 1 && plan[0] == 0 && limit - frozen >= fresh_split_min && !x_t0) {
      // The first round of a block in rank-ordered sub-launches, each behind a pass that settles the walks whose seed is claimed by
      // then (ext_plan_kernel): the heaviest seeds of the block run first and take their whole transcripts; of the seeds behind them
      // -- 98.6 % of the walks of BASELINE configs[2] end void, and a transcript's k1-mers have similar weights, so they sit in the same
      // block as the walk that swallows them -- only the ones still free are launched at all.  The sub-ranges double (1/2^(S-1) of the
      // block first).  Semantically this is the same round under one of its possible schedules: a walk sees the live claims of lower
      // ranks, a seed found claimed by a lower rank is void exactly as when its own thread finds it so (and is marked again if that
      // claim is given up later: seed_rank in the mark pass).
      const unsigned long long total = limit - frozen;
      uint32_t a = frozen;
      for (uint32_t j = 0; j < fresh_split && a < limit; j++) {
        const uint32_t b = j + 1 == fresh_split ? limit : (uint32_t)std::min<unsigned long long>(limit, (unsigned long long)frozen + std::max<unsigned long long>(16, total >> (fresh_split - 1 - j)));
        if (b <= a) continue;
        TRYE(hipMemsetAsync(d_cnt + 2, 0, 32, s));
        hipLaunchKernelGGL(ext_plan_kernel, dim3((uint32_t)cdiv(b - a, 1024)), dim3(1024), 0, s, e->d_nr, e->d_nl, (uint64_t)b, a,
                           mvalid, mR, mL, dirty, long_list, short_list, d_cnt + 2, 0xFFFFFFFFu, coarse,
                           ((a > 0) && prepass) ? (const u64*)claim : (const u64*)nullptr, e->d_order, e->d_totw,
                           (const Rec*)e->d_rec, settle_hops, settled, d_cnt + 20);
        TRYE(hipMemcpyAsync(plan, d_cnt + 2, 32, hipMemcpyDeviceToHost, s));
        TRYE(hipStreamSynchronize(s));
        if (plan[2]) {
          TimerRegion tk(ctx, T_EXT_WALK_FRESH);
          hipLaunchKernelGGL(ext_walk_kernel<true>, dim3((uint32_t)cdiv(plan[2], WBLK)), dim3(WBLK), 0, s, A, (uint64_t)plan[2], short_list, snap);
        }
        a = b;
      }
      plan[0] = 0; plan[2] = 0;                   // (nothing handed over, nothing for the wavefront kernels in this round)
    } else
    if (plan[2]) {
      TimerRegion tk(ctx, fresh_block ? T_EXT_WALK_FRESH : T_EXT_WALK_THREAD);
      if (fresh_block) hipLaunchKernelGGL(ext_walk_kernel<true>, dim3((uint32_t)cdiv(plan[2], WBLK)), dim3(WBLK), 0, s, A, (uint64_t)plan[2], short_list, snap);
      else hipLaunchKernelGGL(ext_walk_kernel<false>, dim3((uint32_t)cdiv(plan[2], WBLK)), dim3(WBLK), 0, s, A, (uint64_t)plan[2], short_list, snap);
    }
    if (x_t0 > 0) {
      TRYE(hipStreamSynchronize(s)); timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
      unsigned long long st = 0, lw = 0; TRYE(hipMemcpyAsync(&st, d_cnt + 1, 8, hipMemcpyDeviceToHost, s));
      if (A.dbg) { TRYE(hipMemcpyAsync(&lw, d_cnt + 44, 8, hipMemcpyDeviceToHost, s)); TRYE(hipMemsetAsync(d_cnt + 44, 0, 8, s)); }
      TRYE(hipStreamSynchronize(s));
      unsigned long long nrs[3] = {0, 0, 0}; TRYE(hipMemcpyAsync(nrs, d_cnt + 21, 24, hipMemcpyDeviceToHost, s)); TRYE(hipMemsetAsync(d_cnt + 21, 0, 24, s)); TRYE(hipStreamSynchronize(s));
      fprintf(stderr, "[shn_extend] XTIME round %d [%u,%u): of the claim holders without a memo %llu were robbed while they sat out; their walks: longest %llu steps, %llu steps in all\n", it + 1, frozen, limit, nrs[0], nrs[1], nrs[2]);
      fprintf(stderr, "[shn_extend] XTIME round %d: release %s (claims to give back by the records: %llu; bulk %d dense %d fresh %d snapshot current %d)\n", it + 1,
              was_fresh ? "none (a new block)" : memo_release_done ? "through memos / logs" : "the begin pass", plan[7], (int)bulk, (int)dense, (int)was_fresh, (int)snap_was_current);
      fprintf(stderr, "[shn_extend] XTIME round %d: %llu dirty walks, %llu of them hold claims without a current memo (rounds released through memos so far: %d); thread walker %llu walks, %.2f ms, steps so far %llu; longest walk %llu steps in %.2f ms (%.2f us per step)\n", it + 1, plan[3], plan[1], n_begin_skipped, plan[2],
              ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6 - x_t0, st, lw >> 32, (double)(lw & 0xFFFFFFFFULL) * 1e-5, (lw >> 32) ? (double)(lw & 0xFFFFFFFFULL) * 1e-2 / (double)(lw >> 32) : 0.0);
    }
    if (plan[2] && bulk) {                      // ... of a bulk round: packed, a lane per walk, lanes refilled from the list (ext_walk_resume_kernel)
      if (bulk_promote) {
        TimerRegion tk(ctx, T_EXT_WALK_THREAD);
        const uint32_t rgrid = (uint32_t)std::min<unsigned long long>(cdiv(plan[2], WBLK), resume_waves);
        if (was_fresh) hipLaunchKernelGGL(ext_walk_resume_kernel<true>, dim3(rgrid), dim3(WBLK), 0, s, A, promo_list, (const unsigned long long*)(d_cnt + 13), d_cnt + 14, snap);
        else hipLaunchKernelGGL(ext_walk_resume_kernel<false>, dim3(rgrid), dim3(WBLK), 0, s, A, promo_list, (const unsigned long long*)(d_cnt + 13), d_cnt + 14, snap);
      }
    } else if (plan[2]) {                       // walks the thread kernel handed over (the count stays on the device)
      TimerRegion tk(ctx, T_EXT_WALK_WAVE);
      hipLaunchKernelGGL(ext_walk_long_kernel<true>, dim3((uint32_t)std::min<unsigned long long>(plan[2], 8192ULL)), dim3(64), 0, s, A, promo_list,
                         (uint64_t)ns, (const unsigned long long*)(d_cnt + 13));
    }
    if (plan[0]) TRYE(hipStreamWaitEvent(s, ev_join, 0));
    // A block's first round when it is a bulk round: no walk of the block held a claim before it, so every change is
    // "nobody -> a walk of the block" -- nothing the mark pass would mark (a k1-mer only BECOMES available to somebody when a lower
    // rank gives it up), no memo slots to fill; what is left of the pass is bringing the snapshot up to date, which the next begin
    // pass does while it streams the claims anyway (copy = 1): the pass is skipped (3 x 25 ms at BASELINE configs[2]).
    const bool skip_mark = fresh_block && bulk && precise_marks;
    fresh_block = false;
    if (dense) e->dense_rounds++;
    // who has to run next round?  walks whose view changed (mark) + walks that lost a claim race (verify);
    // the walks that ran get their memo rebuilt from the claims
    hipLaunchKernelGGL(ext_memo_plan_kernel, dim3((uint32_t)cdiv(limit - frozen, 256)), dim3(256), 0, s, ran, dirty, e->d_nr, e->d_nl, e->d_order, frozen, limit,
                       moff, mR, mL, mvalid, fill, pool, d_cnt + 10, pool_cap, bulk ? 0xFFFFFFFFu : memo_min, use_logs ? log_head : (uint32_t*)nullptr);
    if (skip_mark) { snap_current = false; if (!dense) TRYE(hipMemsetAsync(chunk, 0, n_chunks, s)); }
    else
    { TimerRegion tk(ctx, T_EXT_MARK);
      hipLaunchKernelGGL(ext_mark_kernel, dim3(std::min<uint32_t>(g2n, 4096u)), dim3(256), 0, s, claim, snap, 2 * n, e->d_rec,
                         dirty, ran, d_cnt + 6, frozen, limit, bulk ? (const uint8_t*)nullptr : (const uint8_t*)fill, moff, mR, pool, e->d_nr, e->d_nl, precise_marks,
                         dense ? (const uint8_t*)nullptr : chunk, robsat);
      if (!dense) TRYE(hipMemsetAsync(chunk, 0, n_chunks, s)); }
    hipLaunchKernelGGL(ext_verify_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, ran, robbed, (uint64_t)ns, dirty);
    if (getenv("SHN_EXT_FAULT") && it + 1 == atoi(getenv("SHN_EXT_FAULT"))) TRYE(hipMemsetAsync(dirty, 0, ns + 1, s));   // (tests: lose every mark of this round)
    if (getenv("SHN_EXT_ALLDIRTY")) { TRYE(hipMemsetAsync(dirty, 0, ns + 1, s)); TRYE(hipMemsetAsync(dirty + frozen, 1, limit - frozen, s)); }
    it++;

    if (getenv("SHN_DEBUG")) {
      unsigned long long chg = 0, cur = 0, mx[2] = {0, 0}, st_tr[2] = {0, 0};
      TRYE(hipMemcpyAsync(&st_tr[0], d_cnt + 1, 8, hipMemcpyDeviceToHost, s));
      TRYE(hipMemcpyAsync(&st_tr[1], d_cnt + 15, 8, hipMemcpyDeviceToHost, s));
      TRYE(hipMemcpyAsync(&chg, d_cnt + 6, 8, hipMemcpyDeviceToHost, s));
      TRYE(hipMemcpyAsync(&cur, d_cnt + 10, 8, hipMemcpyDeviceToHost, s));
      TRYE(hipMemcpyAsync(mx, d_cnt + 42, 16, hipMemcpyDeviceToHost, s));
      TRYE(hipMemsetAsync(d_cnt + 42, 0, 16, s));
      TRYE(hipStreamSynchronize(s));
      static double t_prev = 0;
      timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
      double tn = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
      static unsigned long long st_prev[2] = {0, 0};
      if (it == 1) { st_prev[0] = st_prev[1] = 0; }
      fprintf(stderr, "[shn_extend] round %d [%u,%u): dirty=%llu long=%llu short=%llu changed_kmers=%llu pool=%.1f%% longest wavefront walk: %llu steps, most sequential: %llu (no hint %llu, owner without memo %llu, memo moved on %llu, followed %llu)  thread steps %llu in %llu wavefront trips (lanes busy %.3f)  %.2f ms\n", it, frozen, limit,
              plan[3], plan[0], plan[2], chg, 100.0 * (double)cur / (double)pool_cap, mx[1], mx[0] >> 48, (mx[0] >> 36) & 4095, (mx[0] >> 24) & 4095, (mx[0] >> 12) & 4095, mx[0] & 4095,
              st_tr[0] - st_prev[0], st_tr[1] - st_prev[1], (double)(st_tr[0] - st_prev[0]) / (64.0 * (double)std::max<unsigned long long>(1, st_tr[1] - st_prev[1])), it == 1 ? 0.0 : tn - t_prev);
      st_prev[0] = st_tr[0]; st_prev[1] = st_tr[1];
      t_prev = tn;
    }
  }
  hipStreamSynchronize(aux);
  hipStreamDestroy(aux);
  hipEventDestroy(ev_fork);
  hipEventDestroy(ev_join);
  if (converged && ns && getenv("SHN_EXT_AUDIT")) {
    WalkArgs A;
    memset(&A, 0, sizeof(A));
    A.order = e->d_order; A.adjR = rows_R(e->d_rec); A.adjL = rows_L(e->d_rec); A.weight = words_weight(e->d_rec);
    A.claim = claim; A.nr_out = e->d_nr; A.nl_out = e->d_nl; A.totw_out = e->d_totw;
    unsigned long long au[2] = {0, ~0ULL};
    plan[4] = 0; plan[5] = ~0ULL;
    TRYE(hipMemcpyAsync(d_cnt + 48, plan + 4, 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ext_audit_kernel, dim3((uint32_t)cdiv(ns, 64)), dim3(64), 0, s, A, (uint64_t)ns, d_cnt + 48);
    TRYE(hipMemcpyAsync(plan + 4, d_cnt + 48, 16, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    au[0] = plan[4]; au[1] = plan[5];
    if (au[0]) {
      fprintf(stderr, "[shn_extend] AUDIT: %llu walks are not at their fixpoint, lowest rank %llu of %llu (rounds %d)\n", au[0], au[1], ns, it);
      if (atoi(getenv("SHN_EXT_AUDIT")) > 1) { shn_ext_destroy(e); return shn_fail(SHN_ERR_INTERNAL, "shn_extend: audit failed"); }
    }
  }
  e->iterations = it;
  if (!converged) { shn_ext_destroy(e); return shn_fail(SHN_ERR_INTERNAL, "shn_extend: walk fixpoint did not converge"); }
  if (want_dig) {
    int rd;
    if ((rd = ext_digest(ctx, e, 6, claim, 2 * n * 8, 8)) || (rd = ext_digest(ctx, e, 7, e->d_nr, ns * 4, 9)) || (rd = ext_digest(ctx, e, 7, e->d_nl, ns * 4, 10)) ||
        (rd = ext_digest(ctx, e, 7, e->d_totw, ns * 8, 11))) { shn_ext_destroy(e); return rd; }
  }
  // what is left to do with the result (stats, emit, seed info, weights) reads the claims, the walk records and the table: the
  // records and the snapshot (most of the state) go back to the allocator now
  TRYE(hipStreamSynchronize(s));
  shn_dev_free(e->d_rec); e->d_rec = nullptr;
  e->d_claim2 = nullptr;                                   // (not used any more; its memory goes back with the claims')
  unsigned long long steps = 0, wsteps = 0, fsteps = 0, wslots[64];
  TRYE(hipMemcpyAsync(&steps, d_cnt + 1, 8, hipMemcpyDeviceToHost, s));          // thread-kernel steps
  TRYE(hipMemcpyAsync(&fsteps, d_cnt + 16, 8, hipMemcpyDeviceToHost, s));
  { unsigned long long nset = 0; TRYE(hipMemcpyAsync(&nset, d_cnt + 20, 8, hipMemcpyDeviceToHost, s)); TRYE(hipStreamSynchronize(s)); e->settled_walks = nset; }
  TRYE(hipMemcpyAsync(wslots, d_cnt + 64, 64 * 8, hipMemcpyDeviceToHost, s));    // wavefront-kernel steps
  TRYE(hipStreamSynchronize(s));
  for (int i = 0; i < 64; i++) wsteps += wslots[i];
  steps += wsteps;
  if (getenv("SHN_DEBUG")) {
    unsigned long long dbg[10];
    TRYE(hipMemcpyAsync(dbg, d_cnt + 32, 80, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    fprintf(stderr, "[shn_extend] converged after %d rounds; steps: %llu total, %llu in the wave kernel (%llu from own memos, %llu from foreign memos)\n",
            it, steps, wsteps, dbg[0], dbg[1]);
    fprintf(stderr, "[shn_extend] memo_follow: no hint %llu, owner without memo %llu, memo moved on %llu, followed backwards %llu, nothing left %llu, followed %llu; "
            "chunks broken at the first step %llu, later %llu\n", dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9]);
  }
  e->total_steps = steps;
  e->wave_steps = wsteps;
  e->fresh_steps = fsteps;
  TRYE(hipGetLastError());
#undef TRYE
  *out = e;
  return SHN_OK;
}

extern "C" uint64_t shn_ext_n_walks(const shn_ext* e) { return e ? e->n_seeds : 0; }
extern "C" int shn_ext_iterations(const shn_ext* e) { return e ? e->iterations : 0; }
extern "C" uint64_t shn_ext_total_steps(const shn_ext* e) { return e ? e->total_steps : 0; }
extern "C" uint64_t shn_ext_wave_steps(const shn_ext* e) { return e ? e->wave_steps : 0; }
extern "C" uint64_t shn_ext_fresh_steps(const shn_ext* e) { return e ? e->fresh_steps : 0; }
extern "C" int shn_ext_dense_rounds(const shn_ext* e) { return e ? e->dense_rounds : 0; }
extern "C" uint64_t shn_ext_settled_walks(const shn_ext* e) { return e ? e->settled_walks : 0; }

extern "C" int shn_ext_stats_range(shn_ctx* ctx, const shn_ext* e, uint64_t lo, uint64_t n, uint32_t* n_right, uint32_t* n_left, uint64_t* tot_weight) {
  if (!ctx || !e || (n && (!n_right || !n_left || !tot_weight))) return shn_fail(SHN_ERR_ARG, "shn_ext_stats_range: NULL argument");
  if (lo + n > e->n_seeds) return shn_fail(SHN_ERR_ARG, "shn_ext_stats_range: range outside the walks");
  if (!n) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  HIP_TRY(hipMemcpyAsync(n_right, e->d_nr + lo, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(n_left, e->d_nl + lo, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(tot_weight, e->d_totw + lo, n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SHN_OK;
}

extern "C" int shn_ext_stats(shn_ctx* ctx, const shn_ext* e, uint32_t* n_right, uint32_t* n_left, uint64_t* tot_weight) {
  if (!ctx || !e) return shn_fail(SHN_ERR_ARG, "shn_ext_stats: NULL argument");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  if (n_right) HIP_TRY(hipMemcpyAsync(n_right, e->d_nr, e->n_seeds * 4, hipMemcpyDeviceToHost, s));
  if (n_left) HIP_TRY(hipMemcpyAsync(n_left, e->d_nl, e->n_seeds * 4, hipMemcpyDeviceToHost, s));
  if (tot_weight) HIP_TRY(hipMemcpyAsync(tot_weight, e->d_totw, e->n_seeds * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SHN_OK;
}

// ---- the non-void walks only (a few percent of the seeds), in seed order: what the accept filter needs
__global__ void ext_live_flag_kernel(const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, uint64_t ns, uint32_t min_steps,
                                     uint32_t* __restrict__ flag) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < ns) { const uint32_t a = nr[r]; flag[r] = (a != UNCLAIMED && (uint64_t)a + nl[r] >= min_steps) ? 1u : 0u; }
}
__global__ void ext_live_gather_kernel(const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, const uint64_t* __restrict__ totw,
                                       const uint64_t* __restrict__ pos, uint64_t ns, uint32_t* __restrict__ o_rank,
                                       uint32_t* __restrict__ o_nr, uint32_t* __restrict__ o_nl, uint64_t* __restrict__ o_tw) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ns) return;
  uint64_t p = pos[r];
  if (pos[r + 1] == p) return;                                  // not selected by the flag pass
  o_rank[p] = (uint32_t)r; o_nr[p] = nr[r]; o_nl[p] = nl[r]; o_tw[p] = totw[r];
}

extern "C" int shn_ext_live_stats_min(shn_ctx* ctx, const shn_ext* e, uint32_t min_steps, uint64_t* n_live, uint32_t* rank, uint32_t* n_right,
                                      uint32_t* n_left, uint64_t* tot_weight);
extern "C" int shn_ext_live_stats(shn_ctx* ctx, const shn_ext* e, uint64_t* n_live, uint32_t* rank, uint32_t* n_right, uint32_t* n_left,
                                  uint64_t* tot_weight) {
  return shn_ext_live_stats_min(ctx, e, 0, n_live, rank, n_right, n_left, tot_weight);
}
// ... of the non-void walks of at least min_steps steps (the first clause of the accept filter, extension_correction.py:361, is a
// bound on the contig length k1 + steps: at BASELINE configs[2] it leaves 0.7 M of tens of millions of live walks to download)
extern "C" int shn_ext_live_stats_min(shn_ctx* ctx, const shn_ext* e, uint32_t min_steps, uint64_t* n_live, uint32_t* rank, uint32_t* n_right,
                                      uint32_t* n_left, uint64_t* tot_weight) {
  if (!ctx || !e || !n_live) return shn_fail(SHN_ERR_ARG, "shn_ext_live_stats: NULL argument");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t ns = e->n_seeds;
  if (!ns) { *n_live = 0; return SHN_OK; }
  void *pf, *pp, *po;
  int rc;
  if ((rc = shn_ws(ctx)[9].get((ns + 1) * 4, &pf)) || (rc = shn_ws(ctx)[11].get((ns + 2) * 8, &pp))) return rc;
  uint32_t* flag = (uint32_t*)pf;
  uint64_t* pos = (uint64_t*)pp;
  hipLaunchKernelGGL(ext_live_flag_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, e->d_nr, e->d_nl, ns, min_steps, flag);
  uint64_t total = 0;
  if ((rc = shn_device_scan_u32(ctx, flag, ns, pos, &total))) return rc;
  if (!rank) { *n_live = total; return SHN_OK; }                  // sizing call
  if (*n_live < total) return shn_fail(SHN_ERR_ARG, "shn_ext_live_stats: output arrays too small");
  *n_live = total;
  if (!total) return SHN_OK;
  if ((rc = shn_ws(ctx)[10].get(total * 20 + 64, &po))) return rc;
  uint64_t* o_tw = (uint64_t*)po;
  uint32_t* o_rank = (uint32_t*)(o_tw + total);
  uint32_t* o_nr = o_rank + total;
  uint32_t* o_nl = o_nr + total;
  hipLaunchKernelGGL(ext_live_gather_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, e->d_nr, e->d_nl, e->d_totw, pos, ns, o_rank, o_nr,
                     o_nl, o_tw);
  HIP_TRY(hipMemcpyAsync(rank, o_rank, total * 4, hipMemcpyDeviceToHost, s));
  if (n_right) HIP_TRY(hipMemcpyAsync(n_right, o_nr, total * 4, hipMemcpyDeviceToHost, s));
  if (n_left) HIP_TRY(hipMemcpyAsync(n_left, o_nl, total * 4, hipMemcpyDeviceToHost, s));
  if (tot_weight) HIP_TRY(hipMemcpyAsync(tot_weight, o_tw, total * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

// ---- the accept filter itself (extension_correction.py:361: len >= min_length and len * avg_weight ** 0.25 >= threshold) over the
// non-void walks, in seed order: class 1 = passes for sure, 2 = within 1e-9 (relative) of the threshold -- the caller decides those
// few with the reference's own arithmetic (math.pow); two square roots here stand for the fourth root, a few ulp from pow.
__global__ void ext_accept_flag_kernel(const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, const uint64_t* __restrict__ totw, uint64_t ns,
                                       int k, uint32_t min_length, double thr, uint32_t* __restrict__ flag) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ns) return;
  const uint32_t a = nr[r];
  uint32_t f = 0;
  if (a != UNCLAIMED) {
    const uint64_t steps = (uint64_t)a + nl[r], len = steps + (uint64_t)k;
    if (len >= min_length) {
      const double avg = (double)totw[r] / (double)(steps + 1);
      const double lhs = (double)len * sqrt(sqrt(avg));
      f = lhs >= thr * (1.0 - 1e-9) ? 1u : 0u;
    }
  }
  flag[r] = f;
}
__global__ void ext_accept_gather_kernel(const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, const uint64_t* __restrict__ totw,
                                         const uint64_t* __restrict__ pos, uint64_t ns, int k, double thr, uint32_t* __restrict__ o_rank,
                                         uint32_t* __restrict__ o_steps, uint64_t* __restrict__ o_tw, uint8_t* __restrict__ o_cls) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ns) return;
  const uint64_t p = pos[r];
  if (pos[r + 1] == p) return;
  const uint64_t steps = (uint64_t)nr[r] + nl[r];
  const double lhs = (double)(steps + (uint64_t)k) * sqrt(sqrt((double)totw[r] / (double)(steps + 1)));
  o_rank[p] = (uint32_t)r; o_steps[p] = (uint32_t)steps; o_tw[p] = totw[r];
  o_cls[p] = lhs >= thr * (1.0 + 1e-9) ? 1 : 2;
}
// n_out: in = room of the output arrays (0 with NULL arrays: a sizing call), out = candidates; rank / steps (= n_right + n_left) /
// tot_weight / cls per candidate, in seed order
extern "C" int shn_ext_accept(shn_ctx* ctx, const shn_ext* e, uint32_t min_length, double threshold, uint64_t* n_out, uint32_t* rank, uint32_t* steps,
                              uint64_t* tot_weight, uint8_t* cls) {
  if (!ctx || !e || !n_out) return shn_fail(SHN_ERR_ARG, "shn_ext_accept: NULL argument");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const uint64_t ns = e->n_seeds;
  if (!ns) { *n_out = 0; return SHN_OK; }
  void *pf, *pp, *po;
  int rc;
  if ((rc = shn_ws(ctx)[9].get((ns + 1) * 4, &pf)) || (rc = shn_ws(ctx)[11].get((ns + 2) * 8, &pp))) return rc;
  uint32_t* flag = (uint32_t*)pf;
  uint64_t* pos = (uint64_t*)pp;
  hipLaunchKernelGGL(ext_accept_flag_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, e->d_nr, e->d_nl, e->d_totw, ns, e->k, min_length, threshold, flag);
  uint64_t total = 0;
  if ((rc = shn_device_scan_u32(ctx, flag, ns, pos, &total))) return rc;
  if (!rank) { *n_out = total; return SHN_OK; }
  if (*n_out < total || !steps || !tot_weight || !cls) return shn_fail(SHN_ERR_ARG, "shn_ext_accept: output arrays too small");
  *n_out = total;
  if (!total) return SHN_OK;
  if ((rc = shn_ws(ctx)[10].get(total * 17 + 64, &po))) return rc;
  uint64_t* o_tw = (uint64_t*)po;
  uint32_t* o_rank = (uint32_t*)(o_tw + total);
  uint32_t* o_steps = o_rank + total;
  uint8_t* o_cls = (uint8_t*)(o_steps + total);
  hipLaunchKernelGGL(ext_accept_gather_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, e->d_nr, e->d_nl, e->d_totw, pos, ns, e->k, threshold, o_rank,
                     o_steps, o_tw, o_cls);
  HIP_TRY(hipMemcpyAsync(rank, o_rank, total * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(steps, o_steps, total * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(tot_weight, o_tw, total * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(cls, o_cls, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

// seed string (oriented k1-mer key) and seed weight of the given walks: the global order of the walks is
// (weight descending, key ascending), which is what merges the candidates of several shards
__global__ void ext_seed_info_kernel(const uint32_t* __restrict__ ranks, uint64_t n, const uint32_t* __restrict__ order,
                                     const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ weight, int k,
                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ w) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t o = order[ranks[j]];
  keys[j] = oriented_string(tkeys, o, k);
  w[j] = weight[o >> 1];
}
extern "C" int shn_ext_seed_info(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n, uint64_t* keys, uint32_t* weights) {
  if (!ctx || !e || (n && (!ranks || !keys || !weights))) return shn_fail(SHN_ERR_ARG, "shn_ext_seed_info: NULL argument");
  if (!n) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  for (uint64_t j = 0; j < n; j++) if (ranks[j] >= e->n_seeds) return shn_fail(SHN_ERR_ARG, "shn_ext_seed_info: rank out of range");
  uint32_t *dr, *dw; uint64_t* dk;
  HIP_TRY(shn_dev_malloc(&dr, n * 4)); HIP_TRY(shn_dev_malloc(&dw, n * 4)); HIP_TRY(shn_dev_malloc(&dk, n * 8));
  HIP_TRY(hipMemcpyAsync(dr, ranks, n * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ext_seed_info_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, dr, n, e->d_order, e->table->d_keys, e->d_weight, e->k, dk, dw);
  HIP_TRY(hipMemcpyAsync(keys, dk, n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(weights, dw, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  shn_dev_free(dr); shn_dev_free(dw); shn_dev_free(dk);
  return SHN_OK;
}

// bases_out: the contigs' text on the host; dev_out (instead): the text stays on the device (total + 64 bytes, the tail zeroed; the
// caller frees it with shn_dev_free) -- the GPU contig stage reads it there
static int ext_emit_impl(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n_sel, const uint64_t* offsets, uint8_t* bases_out, uint8_t** dev_out) {
  if (!ctx || !e || (n_sel && (!ranks || !offsets || (!bases_out && !dev_out)))) return shn_fail(SHN_ERR_ARG, "shn_ext_emit: NULL argument");
  if (dev_out) *dev_out = nullptr;
  if (!n_sel) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_EXTEND);
  const uint64_t total = offsets[n_sel], ns = e->n_seeds;
  // rank -> index in the selection (built on the device: the map has one entry per walk, the selection is small);
  // expected number of k1-mers of the selected walks
  for (uint64_t t = 0; t < n_sel; t++)
    if (ranks[t] >= ns) return shn_fail(SHN_ERR_ARG, "shn_ext_emit: rank out of range");
  unsigned long long expect = 0;
  for (uint64_t t = 0; t < n_sel; t++) {
    uint64_t len = offsets[t + 1] - offsets[t];
    if (len < (uint64_t)e->k) return shn_fail(SHN_ERR_ARG, "shn_ext_emit: offsets do not fit the walk lengths");
    expect += len - e->k + 1;
  }
  int32_t* d_sel; uint64_t* d_off; uint8_t* d_out; unsigned long long* d_cnt; uint32_t* d_ranks;
  HIP_TRY(shn_dev_malloc(&d_sel, (ns + 1) * 4));
  HIP_TRY(shn_dev_malloc(&d_off, (n_sel + 1) * 8));
  HIP_TRY(shn_dev_malloc(&d_out, total + 64));
  HIP_TRY(shn_dev_malloc(&d_cnt, 32));
  HIP_TRY(shn_dev_malloc(&d_ranks, (n_sel + 1) * 4));
  HIP_TRY(hipMemsetAsync(d_sel, 0xFF, (ns + 1) * 4, s));                    // -1: not selected
  HIP_TRY(hipMemcpyAsync(d_ranks, ranks, n_sel * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_off, offsets, (n_sel + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 32, s));
  hipLaunchKernelGGL(ext_select_kernel, dim3((uint32_t)cdiv(n_sel, 256)), dim3(256), 0, s, d_ranks, n_sel, d_sel, d_cnt + 2);
  HIP_TRY(hipMemsetAsync(d_out, 0, total + 64, s));
  {
    TimerRegion tk(ctx, T_EXT_EMIT);
    hipLaunchKernelGGL(ext_emit_claims_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(2 * e->n, 256), 4096)), dim3(256), 0, s, e->d_claim, 2 * e->n, d_sel, ns,
                       e->d_nr, e->d_nl, e->table->d_keys, e->k, d_off, d_out, d_cnt);
  }
  unsigned long long cnt[3] = {0, 0, 0};
  if (bases_out) HIP_TRY(hipMemcpyAsync(bases_out, d_out, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(cnt, d_cnt, 24, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  shn_dev_free(d_sel); shn_dev_free(d_off); shn_dev_free(d_cnt); shn_dev_free(d_ranks);
  struct FreeOut { uint8_t* p; ~FreeOut() { if (p) shn_dev_free(p); } } free_out{d_out};
  HIP_TRY(hipGetLastError());
  if (cnt[2]) return shn_fail(SHN_ERR_ARG, "shn_ext_emit: a walk is selected twice");
  // every base of every selected contig must have been written exactly once
  if (cnt[1] || cnt[0] != expect)
    return shn_fail(SHN_ERR_INTERNAL, "shn_ext_emit: claims do not match the recorded walks (k1-mers written " + std::to_string(cnt[0]) +
                    ", expected " + std::to_string(expect) + ", stray claims " + std::to_string(cnt[1]) + ")");
  if (dev_out) { *dev_out = d_out; free_out.p = nullptr; }
  return SHN_OK;
}
extern "C" int shn_ext_emit(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n_sel, const uint64_t* offsets,
                            uint8_t* bases_out) {
  if (n_sel && !bases_out) return shn_fail(SHN_ERR_ARG, "shn_ext_emit: NULL argument");
  return ext_emit_impl(ctx, e, ranks, n_sel, offsets, bases_out, nullptr);
}
// the same with the text left on the device (shn_devtext: what shn_contig_stage_device reads; shn_devtext_segments fetches pieces)
extern "C" int shn_ext_emit_device(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n_sel, const uint64_t* offsets, shn_devtext** out) {
  if (!out) return shn_fail(SHN_ERR_ARG, "shn_ext_emit_device: NULL argument");
  *out = nullptr;
  uint8_t* d = nullptr;
  int rc = ext_emit_impl(ctx, e, ranks, n_sel, offsets, nullptr, &d);
  if (rc) return rc;
  shn_devtext* t = new shn_devtext();
  t->ctx = ctx; t->d = d; t->n = n_sel ? offsets[n_sel] : 0;
  *out = t;
  return SHN_OK;
}

// weights of arbitrary k1-mer strings in the doubled input (for the `allowed` dict, :404-408)
__global__ void ext_weight_lookup_kernel(const TabIdx T,
                                         const uint32_t* __restrict__ weight, const uint8_t* __restrict__ flags, int k, int canonical,
                                         const uint64_t* __restrict__ q, uint64_t nq, uint32_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  uint64_t key = q[i];
  if (canonical) { uint64_t rc = shn_revcomp(key, k); key = rc < key ? rc : key; }
  int64_t j = shn_tab_find(T, key);
  out[i] = (j >= 0 && !(flags[j] & 2)) ? weight[j] : 0;
}

extern "C" int shn_ext_weights(shn_ctx* ctx, const shn_ext* e, const uint64_t* keys, uint64_t n, uint32_t* weights) {
  if (!ctx || !e || (n && (!keys || !weights))) return shn_fail(SHN_ERR_ARG, "shn_ext_weights: NULL argument");
  if (e->owned_table) return shn_fail(SHN_ERR_ARG, "shn_ext_weights: not available on a component shard (it holds only this rank's k1-mers)");
  if (!n) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint64_t* dq; uint32_t* dw;
  HIP_TRY(shn_dev_malloc(&dq, n * 8));
  HIP_TRY(shn_dev_malloc(&dw, n * 4));
  HIP_TRY(hipMemcpyAsync(dq, keys, n * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ext_weight_lookup_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, shn_tab_idx(e->table),
                     e->d_weight, e->d_flags, e->k, e->table->canonical, dq, n, dw);
  HIP_TRY(hipMemcpyAsync(weights, dw, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  shn_dev_free(dq); shn_dev_free(dw);
  return SHN_OK;
}
