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
// Two accelerations keep an iteration cheap.  (1) Frozen prefix: the lowest rank whose path changed in an
// iteration, and every rank below it, is final and is never recomputed.  (2) Path memo: a long walk keeps
// the path of its previous iteration; a wavefront re-checks 64 consecutive old steps at once (each lane
// re-decides one step against the current claims) and only walks sequentially from the first step whose
// decision changed until the new path rejoins the old one -- so an unchanged 4,000-step walk costs ~60
// dependent memory round trips per iteration instead of 4,000.
//
// Oriented k1-mers: the count table stores canonical keys; oriented id o = 2*i + s is the string
// key_i (s=0) or its reverse complement (s=1; unused for palindromes).  Both strands are walked,
// as in the reference's strand-doubled input.
#include "common.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define EBLK 256
#define UNCLAIMED 0xFFFFFFFFu
#define UNCLAIMED64 0xFFFFFFFFFFFFFFFFULL
#define LONG_WALK 96          // walks at least this long (previous iteration) get a wavefront + path memo
#define POOL_SLACK 256
typedef unsigned long long u64;
#define CLAIM(rank, pos) (((u64)(rank) << 32) | (u64)(uint32_t)(pos))
#define RANK(c) ((uint32_t)((c) >> 32))
#define POS(c) ((uint32_t)(c))

struct shn_ext {
  shn_ctx* ctx;
  int device;
  int k;
  uint64_t n;            // canonical entries
  uint64_t n_seeds;
  int iterations;
  uint32_t min_weight;
  const shn_table* table;
  uint32_t* d_weight;    // [n] weight of the string in the doubled input (count, x2 for palindromes)
  uint8_t* d_flags;      // [n] bit0 palindrome, bit1 low complexity
  int32_t* d_adjR;       // [2n*4] oriented id reached by appending base b, or -1
  int32_t* d_adjL;       // [2n*4] oriented id reached by prepending base b, or -1
  uint32_t* d_order;     // [n_seeds] oriented id of the seed with rank r
  u64* d_claim;          // [2n] converged claims: (rank of the owning walk) << 32 | (1 + step index on its path)
  u64* d_claim2;         // [2n] scratch
  uint32_t* d_pool;      // stored paths of the long walks (converged iteration)
  uint64_t* d_poff;      // [n_seeds] pool offset of walk r (valid when d_pstored[r])
  uint8_t* d_pstored;    // [n_seeds]
  uint64_t total_steps;  // walk steps executed over all iterations (for the bench's byte model)
  uint32_t* d_nr;        // [n_seeds] right steps (UNCLAIMED = void walk)
  uint32_t* d_nl;        // [n_seeds]
  uint64_t* d_totw;      // [n_seeds] sum of weights incl. the seed
  uint64_t* d_hash;      // [n_seeds] path hash of the last iteration
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

__global__ void ext_adjacency_kernel(const uint64_t* __restrict__ tkeys, const uint64_t* __restrict__ boff, int bits,
                                     const uint8_t* __restrict__ flags, uint64_t n, int k, int canonical,
                                     int32_t* __restrict__ adjR, int32_t* __restrict__ adjL) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (oriented, dir, base)
  if (gid >= n * 16) return;
  uint32_t b = gid & 3;
  uint32_t dir = (gid >> 2) & 1;
  uint64_t o = gid >> 3;
  uint64_t i = o >> 1;
  int32_t res = -1;
  uint8_t f = flags[i];
  bool dead = (f & 2) || ((o & 1) && ((f & 1) || !canonical));
  if (!dead) {
    uint64_t str = (o & 1) ? shn_revcomp(tkeys[i], k) : tkeys[i];
    uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t nb = dir == 0 ? (((str << 2) | b) & mask) : ((str >> 2) | ((uint64_t)b << (2 * (k - 1))));
    uint64_t canon = nb;
    uint32_t strand = 0;
    if (canonical) { uint64_t rc = shn_revcomp(nb, k); if (rc < nb) { canon = rc; strand = 1; } }
    int64_t j = shn_table_find(tkeys, boff, bits, canon);
    if (j >= 0 && !(flags[j] & 2)) res = (int32_t)(2 * j + strand);
  }
  (dir == 0 ? adjR : adjL)[o * 4 + b] = res;
}

__global__ void ext_seed_kernel(const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ weight,
                                const uint8_t* __restrict__ flags, uint64_t n, int k, int canonical, uint32_t min_weight,
                                uint64_t* __restrict__ skeys, uint32_t* __restrict__ svals, unsigned long long* __restrict__ counter) {
  uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= 2 * n) return;
  uint64_t i = o >> 1;
  uint8_t f = flags[i];
  if (f & 2) return;
  if ((o & 1) && ((f & 1) || !canonical)) return;
  if (weight[i] < min_weight) return;
  unsigned long long p = atomicAdd(counter, 1ULL);
  skeys[p] = (o & 1) ? shn_revcomp(tkeys[i], k) : tkeys[i];
  svals[p] = (uint32_t)o;
}

__global__ void ext_weightkey_kernel(const uint32_t* __restrict__ svals, const uint32_t* __restrict__ weight, uint64_t ns,
                                     uint64_t* __restrict__ wkeys) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  wkeys[i] = (uint64_t)(0xFFFFFFFFu - weight[svals[i] >> 1]);   // ascending sort => weight descending
}

struct Adj4 { int32_t v[4]; };

struct WalkArgs {
  const uint32_t* order; const Adj4* adjR; const Adj4* adjL; const uint32_t* weight;
  const u64* claim_prev; u64* claim_cur;
  uint32_t first;
  uint32_t* nr_out; uint32_t* nl_out; uint64_t* totw_out; uint64_t* hash_io; uint32_t* changed;
  const uint32_t* pool_prev; const uint64_t* poff_prev; const uint8_t* pstored_prev;
  uint32_t* pool_cur; const uint64_t* poff_cur; const uint32_t* pcap_cur; uint8_t* pstored_cur;
  const uint8_t* is_long;
  unsigned long long* steps_counter;
};

__device__ __forceinline__ uint64_t step_hash(uint32_t node, uint32_t pos) {
  return shn_mix64((uint64_t)node + 0x9E3779B97F4A7C15ULL * (uint64_t)pos);
}

// One greedy decision (extension_correction.py:223-237): among the candidates that exist and are not
// traversed pick the heaviest, ties in BASES order A,G,C,T (codes 0,2,1,3; strict >).  Traversed =
// claimed by a lower rank (previous iteration or final), claimed in this iteration by a rank <= r (own
// trail included), or -- while re-checking a stretch of the old path -- own old position in [lo, hi].
__device__ __forceinline__ int decide(const Adj4& cand, uint32_t r, uint32_t lo, uint32_t hi, const u64* __restrict__ claim_prev,
                                      const u64* claim_cur, const uint32_t* __restrict__ weight, uint32_t dummy, uint32_t& bw) {
  u64 cp[4], cc[4];
  uint32_t w[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    uint32_t idx = cand.v[b] < 0 ? dummy : (uint32_t)cand.v[b];
    cp[b] = claim_prev[idx];
    cc[b] = __hip_atomic_load(&claim_cur[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w[b] = weight[idx >> 1];
  }
  int best = -1;
  bw = 0;
#define CONSIDER(b)                                                                                          \
  if (cand.v[b] >= 0) {                                                                                        \
    bool trav = RANK(cp[b]) < r || RANK(cc[b]) <= r || (RANK(cp[b]) == r && POS(cp[b]) >= lo && POS(cp[b]) <= hi); \
    if (!trav && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }                                             \
  }
  CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
  return best;
}

// ---- short walks: one thread per walk, one memory round trip per step (candidate rows prefetched)
template <bool EMIT>
__global__ __launch_bounds__(EBLK) void ext_walk_kernel(WalkArgs A, uint64_t n_walks, const uint32_t* __restrict__ sel,
                                                        const uint8_t* __restrict__ skip, const uint64_t* __restrict__ tkeys, int k,
                                                        const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out_bases) {
  __shared__ unsigned long long blk_steps;
  if (threadIdx.x == 0) blk_steps = 0;
  __syncthreads();
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mysteps = 0;
  if (t < n_walks) {
    const uint32_t r = EMIT ? sel[t] : (uint32_t)t + A.first;
    if (!(skip && skip[r])) {
      const uint32_t o = A.order[r];
      uint32_t nr = 0, nl = 0;
      uint64_t tot = 0, h = 0;
      bool isvoid = RANK(A.claim_prev[o]) < r;
      uint8_t* dst = nullptr;
      uint32_t nl_known = 0;
      if (EMIT && !isvoid) {
        dst = out_bases + out_off[t];
        nl_known = A.nl_out[r];
        uint64_t s = oriented_string(tkeys, o, k);
        for (int j = 0; j < k; j++) dst[nl_known + j] = "ACGT"[(s >> (2 * (k - 1 - j))) & 3];
      }
      if (!isvoid) {
        atomicMin(&A.claim_cur[o], CLAIM(r, 0));
        tot = A.weight[o >> 1];
        uint32_t pos = 0;
        for (int dir = 0; dir < 2; dir++) {
          const Adj4* adj = dir == 0 ? A.adjR : A.adjL;
          uint32_t steps = 0;
          Adj4 cand = adj[o];
          while (true) {
            u64 cp[4], cc[4];
            uint32_t w[4];
            Adj4 nxt[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
              uint32_t idx = cand.v[b] < 0 ? o : (uint32_t)cand.v[b];
              cp[b] = A.claim_prev[idx];
              cc[b] = __hip_atomic_load(&A.claim_cur[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              w[b] = A.weight[idx >> 1];
              nxt[b] = adj[idx];
            }
            int best = -1;
            uint32_t bw = 0;
#define CONSIDER(b) if (cand.v[b] >= 0 && RANK(cp[b]) >= r && RANK(cc[b]) > r && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }
            CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
            if (best < 0) break;
            uint32_t nbest = (uint32_t)cand.v[best];
            pos++;
            __hip_atomic_fetch_min(&A.claim_cur[nbest], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (EMIT) {
              if (dir == 0) dst[nl_known + k + steps] = "ACGT"[best];
              else dst[nl_known - 1 - steps] = "ACGT"[best];
            }
            steps++;
            tot += bw;
            h += step_hash(nbest, pos);
            cand = best == 0 ? nxt[0] : best == 1 ? nxt[1] : best == 2 ? nxt[2] : nxt[3];
          }
          if (dir == 0) nr = steps; else nl = steps;
        }
      }
      if (!EMIT) {
        uint64_t hh = isvoid ? 0ULL : (h | 1ULL);
        if (A.hash_io[r] != hh) { A.hash_io[r] = hh; atomicMin(A.changed, r); }   // lowest rank whose path changed
        A.nr_out[r] = isvoid ? UNCLAIMED : nr;
        A.nl_out[r] = nl;
        A.totw_out[r] = tot;
        A.pstored_cur[r] = 0;
        mysteps = nr + nl;
      }
    }
  }
  if (!EMIT) {
    if (mysteps) atomicAdd(&blk_steps, (unsigned long long)mysteps);
    __syncthreads();
    if (threadIdx.x == 0 && blk_steps) atomicAdd(A.steps_counter, blk_steps);
  }
}

// ---- long walks: one wavefront per walk, old path re-checked 64 steps at a time
#define NONE32 0xFFFFFFFFu
__global__ __launch_bounds__(64) void ext_walk_long_kernel(WalkArgs A, const uint32_t* __restrict__ long_list) {
  const uint32_t r = long_list[blockIdx.x];
  const int lane = threadIdx.x;
  const uint32_t o = A.order[r];
  const bool isvoid = RANK(A.claim_prev[o]) < r;
  // memo layout in the pool: [nR, nL, node of step 0, node of step 1, ...]
  const bool had = A.pstored_prev[r] != 0;
  const uint32_t* oldhdr = A.pool_prev + (had ? A.poff_prev[r] : 0);
  const uint32_t* oldp = oldhdr + 2;
  const uint32_t oldR = had ? oldhdr[0] : 0, oldL = had ? oldhdr[1] : 0;
  uint32_t* newhdr = A.pool_cur + A.poff_cur[r];
  uint32_t* newp = newhdr + 2;
  const uint32_t cap = A.pcap_cur[r] >= 2 ? A.pcap_cur[r] - 2 : 0;
  uint32_t ns = 0, nr_new = 0, seq_steps = 0;
  uint64_t tot = 0, h = 0;
  if (!isvoid) {
    if (lane == 0) atomicMin(&A.claim_cur[o], CLAIM(r, 0));
    tot = A.weight[o >> 1];
    for (int dir = 0; dir < 2; dir++) {
      const Adj4* adj = dir == 0 ? A.adjR : A.adjL;
      const uint32_t ob = dir == 0 ? 0 : oldR, oe = dir == 0 ? oldR : oldR + oldL;
      uint32_t cur = o;
      uint32_t oi = had ? ob : NONE32;          // old step index expected to follow `cur`
      while (true) {
        int32_t taken = -1;                     // node taken by a sequential / deviating step this round
        uint32_t taken_w = 0;
        if (oi != NONE32) {
          const uint32_t nchunk = min(64u, oe - oi);          // real old steps covered by this round
          const uint32_t s = oi + lane;
          const bool active = (uint32_t)lane <= nchunk && s <= oe;   // lane == nchunk: terminal check if the path ends here
          const bool is_term = active && s == oe;
          int32_t chosen = -1;
          uint32_t bw = 0;
          if (active && ((uint32_t)lane < nchunk || is_term)) {
            uint32_t before = lane == 0 ? cur : oldp[s - 1];
            Adj4 cand = adj[before];
            int b = decide(cand, r, oi + 1, s, A.claim_prev, A.claim_cur, A.weight, o, bw);
            chosen = b < 0 ? -1 : cand.v[b];
          }
          const bool checked = active && ((uint32_t)lane < nchunk || is_term);
          const int32_t expect = (checked && !is_term) ? (int32_t)oldp[s] : -1;
          const u64 bad = __ballot(checked && chosen != expect);
          const uint32_t m = bad ? (uint32_t)(__ffsll((long long)bad) - 1) : 64u;
          const uint32_t conf = min(m, nchunk);               // confirmed old steps: lanes [0, conf)
          // publish the confirmed steps
          uint64_t myw = 0, myh = 0;
          if ((uint32_t)lane < conf) {
            uint32_t node = oldp[s];
            uint32_t pos = ns + lane + 1;
            __hip_atomic_fetch_min(&A.claim_cur[node], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ns + lane < cap) newp[ns + lane] = node;
            myw = A.weight[node >> 1];
            myh = step_hash(node, pos);
          }
          for (int off = 32; off > 0; off >>= 1) { myw += __shfl_xor(myw, off, 64); myh += __shfl_xor(myh, off, 64); }
          tot += myw;
          h += myh;
          if (conf > 0) cur = oldp[oi + conf - 1];
          ns += conf;
          oi += conf;
          if (m == 64u) {
            if (nchunk < 64u) break;            // terminal lane agreed: the old path is still complete
            continue;
          }
          // lane m decided differently (a real step, or the terminal check found a continuation)
          taken = __shfl(chosen, (int)m, 64);
          taken_w = __shfl(bw, (int)m, 64);
          if (taken < 0) break;                 // the walk now stops here
        } else {
          Adj4 cand = adj[cur];
          uint32_t bw = 0;
          int b = decide(cand, r, 1, 0, A.claim_prev, A.claim_cur, A.weight, o, bw);
          if (b < 0) break;
          taken = cand.v[b];
          taken_w = bw;
        }
        // take `taken` as the next step (sequential path), then look for a rejoin with the old path
        {
          uint32_t pos = ns + 1;
          if (lane == 0) {
            __hip_atomic_fetch_min(&A.claim_cur[taken], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ns < cap) newp[ns] = (uint32_t)taken;
          }
          tot += taken_w;
          h += step_hash((uint32_t)taken, pos);
          ns++;
          seq_steps++;
          cur = (uint32_t)taken;
          oi = NONE32;
          if (had) {
            u64 cpv = A.claim_prev[taken];
            if (RANK(cpv) == r && POS(cpv) >= 1) {
              uint32_t p = POS(cpv) - 1;
              if (p >= ob && p < oe && oldp[p] == (uint32_t)taken) oi = p + 1;
            }
          }
        }
      }
      if (dir == 0) nr_new = ns;
    }
  }
  bool keep = false;
  if (isvoid) {
    // seed currently traversed by a lower rank: carry the memo forward, the walk may come back to life
    if (had && oldR + oldL <= cap && A.pcap_cur[r] >= 2) {
      for (uint32_t i = lane; i < oldR + oldL; i += 64) newp[i] = oldp[i];
      if (lane == 0) { newhdr[0] = oldR; newhdr[1] = oldL; }
      keep = true;
    }
  } else if (ns <= cap && A.pcap_cur[r] >= 2) {
    if (lane == 0) { newhdr[0] = nr_new; newhdr[1] = ns - nr_new; }
    keep = true;
  }
  if (lane == 0) {
    uint32_t nr = nr_new, nl = ns - nr_new;
    uint64_t hh = isvoid ? 0ULL : (h | 1ULL);
    if (A.hash_io[r] != hh) { A.hash_io[r] = hh; atomicMin(A.changed, r); }
    A.nr_out[r] = isvoid ? UNCLAIMED : nr;
    A.nl_out[r] = nl;
    A.totw_out[r] = tot;
    A.pstored_cur[r] = keep ? 1 : 0;
    if (ns) atomicAdd(A.steps_counter, (unsigned long long)ns);
  }
}

// classify the live walks for the next iteration and lay out the path pool
__global__ void ext_plan_kernel(const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, uint32_t first, uint64_t ns,
                                const uint32_t* __restrict__ pool_prev, const uint64_t* __restrict__ poff_prev,
                                const uint8_t* __restrict__ pstored_prev, uint8_t* __restrict__ is_long,
                                uint32_t* __restrict__ long_list, uint64_t* __restrict__ poff, uint32_t* __restrict__ pcap,
                                unsigned long long* __restrict__ counters, uint64_t pool_cap) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + first;
  if (r >= ns) return;
  uint32_t a = nr[r];
  uint32_t len = a == UNCLAIMED ? 0 : a + nl[r];
  bool memo = pstored_prev[r] != 0;
  if (memo) { const uint32_t* h = pool_prev + poff_prev[r]; len = max(len, h[0] + h[1]); }
  uint8_t lg = 0;
  if (memo || len >= LONG_WALK) {
    uint64_t cap = (uint64_t)len + POOL_SLACK + 2;
    unsigned long long off = atomicAdd(&counters[1], (unsigned long long)cap);
    bool fits = off + cap <= pool_cap;
    unsigned long long idx = atomicAdd(&counters[0], 1ULL);
    long_list[idx] = (uint32_t)r;
    poff[r] = fits ? off : 0;
    pcap[r] = fits ? (uint32_t)cap : 0;
    lg = 1;
  }
  is_long[r] = lg;
}

// copy the stored paths of the walks that just became final ([first, newfirst)) into the final pool
__global__ void ext_keep_final_kernel(const uint32_t* __restrict__ long_list, uint64_t n_long, uint32_t first, uint32_t newfirst,
                                      const uint32_t* __restrict__ pool, const uint64_t* __restrict__ poff,
                                      const uint8_t* __restrict__ pstored, const uint32_t* __restrict__ nr,
                                      uint32_t* __restrict__ fpool, uint64_t* __restrict__ foff, uint8_t* __restrict__ fstored,
                                      unsigned long long* __restrict__ cursor, uint64_t fcap) {
  if (blockIdx.x >= n_long) return;
  uint32_t r = long_list[blockIdx.x];
  if (r < first || r >= newfirst || !pstored[r] || nr[r] == UNCLAIMED) return;
  const uint32_t* h = pool + poff[r];
  uint32_t len = h[0] + h[1];
  __shared__ unsigned long long base;
  if (threadIdx.x == 0) base = atomicAdd(cursor, (unsigned long long)len);
  __syncthreads();
  if (base + len > fcap) return;
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) fpool[base + i] = h[2 + i];
  if (threadIdx.x == 0) { foff[r] = base; fstored[r] = 1; }
}

// Freeze the claims of the walks that just became final into `fin`, and merge every final claim into
// `cur`, which is the next iteration's claim_prev (final claims have rank < first <= any live walk).
__global__ void ext_freeze_kernel(u64* __restrict__ cur, u64* __restrict__ fin, uint64_t n2, uint32_t newfirst) {
  uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n2) return;
  u64 c = cur[o];
  u64 f = fin[o];
  if (RANK(c) < newfirst && c < f) { fin[o] = c; f = c; }
  if (f < c) cur[o] = f;
}

// contig bases of a walk whose path is stored: one block per selected walk
__global__ void ext_emit_stored_kernel(const uint32_t* __restrict__ sel, uint64_t n_sel, const uint32_t* __restrict__ order,
                                       const uint8_t* __restrict__ pstored, const uint32_t* __restrict__ pool,
                                       const uint64_t* __restrict__ poff, const uint32_t* __restrict__ nr_a,
                                       const uint32_t* __restrict__ nl_a, const uint64_t* __restrict__ tkeys, int k,
                                       const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out_bases) {
  uint64_t t = blockIdx.x;
  if (t >= n_sel) return;
  uint32_t r = sel[t];
  if (!pstored[r] || nr_a[r] == UNCLAIMED) return;
  uint32_t nr = nr_a[r], nl = nl_a[r];
  uint8_t* dst = out_bases + out_off[t];
  const uint32_t* p = pool + poff[r];
  uint64_t s = oriented_string(tkeys, order[r], k);
  for (uint32_t j = threadIdx.x; j < (uint32_t)k; j += blockDim.x) dst[nl + j] = "ACGT"[(s >> (2 * (k - 1 - j))) & 3];
  for (uint32_t j = threadIdx.x; j < nr + nl; j += blockDim.x) {
    uint64_t str = oriented_string(tkeys, p[j], k);
    if (j < nr) dst[nl + k + j] = "ACGT"[str & 3];                       // appended base = last base of the k1-mer
    else dst[nl - 1 - (j - nr)] = "ACGT"[(str >> (2 * (k - 1))) & 3];   // prepended base = first base
  }
}

extern "C" void shn_ext_destroy(shn_ext* e) {
  if (!e) return;
  hipSetDevice(e->device);
  void* ptrs[] = {e->d_weight, e->d_flags, e->d_adjR, e->d_adjL, e->d_order, e->d_claim, e->d_claim2, e->d_nr, e->d_nl,
                  e->d_totw, e->d_hash, e->d_pool, e->d_poff, e->d_pstored};
  for (void* p : ptrs) if (p) hipFree(p);
  delete e;
}

extern "C" int shn_extend(shn_ctx* ctx, const shn_table* t, uint32_t min_weight, int max_iterations, shn_ext** out) {
  if (!ctx || !t || !out) return shn_fail(SHN_ERR_ARG, "shn_extend: NULL argument");
  if (2 * t->n >= 0x7FFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_extend: table too large for 31-bit oriented ids");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  TimerRegion treg(ctx, T_EXTEND);
  shn_ext* e = new shn_ext();
  memset(e, 0, sizeof(*e));
  e->ctx = ctx; e->device = ctx->device; e->k = t->k; e->n = t->n; e->min_weight = min_weight; e->table = t;
  uint64_t n = t->n;
  if (max_iterations <= 0) max_iterations = 100000;
#define TRYE(x) do { hipError_t _e = (x); if (_e != hipSuccess) { shn_ext_destroy(e); \
      return shn_fail(SHN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } } while (0)
  TRYE(hipMalloc(&e->d_weight, (n + 1) * 4));
  TRYE(hipMalloc(&e->d_flags, n + 1));
  TRYE(hipMalloc(&e->d_adjR, (2 * n + 1) * 16));
  TRYE(hipMalloc(&e->d_adjL, (2 * n + 1) * 16));
  TRYE(hipMalloc(&e->d_claim, (2 * n + 1) * 8));
  TRYE(hipMalloc(&e->d_claim2, (2 * n + 1) * 8));
  if (n) {
    TimerRegion t1(ctx, T_EXT_PREP);
    hipLaunchKernelGGL(ext_prepare_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, t->d_keys, t->d_counts, n, t->k,
                       t->canonical, e->d_weight, e->d_flags);
    hipLaunchKernelGGL(ext_adjacency_kernel, dim3((uint32_t)cdiv(n * 16, 256)), dim3(256), 0, s, t->d_keys, t->d_bucket_off,
                       t->bits, e->d_flags, n, t->k, t->canonical, e->d_adjR, e->d_adjL);
  }
  // seeds: compact, sort by string then (stable) by weight descending
  void *pk, *pv, *pk2, *pv2, *pc;
  int rc;
  if ((rc = g_shn_ws[9].get((2 * n + 2) * 8, &pk)) || (rc = g_shn_ws[10].get((2 * n + 2) * 4, &pv)) ||
      (rc = g_shn_ws[11].get((2 * n + 2) * 8, &pk2)) || (rc = g_shn_ws[12].get((2 * n + 2) * 4, &pv2)) ||
      (rc = g_shn_ws[13].get(256, &pc))) { shn_ext_destroy(e); return rc; }
  uint64_t* skeys = (uint64_t*)pk; uint32_t* svals = (uint32_t*)pv;
  unsigned long long* d_cnt = (unsigned long long*)pc;      // [0] seed count, [1] steps, [2..3] plan counters
  uint32_t* d_changed = (uint32_t*)(d_cnt + 8);
  TRYE(hipMemsetAsync(d_cnt, 0, 128, s));
  if (n) hipLaunchKernelGGL(ext_seed_kernel, dim3((uint32_t)cdiv(2 * n, 256)), dim3(256), 0, s, t->d_keys, e->d_weight, e->d_flags, n,
                            t->k, t->canonical, min_weight, skeys, svals, d_cnt);
  unsigned long long ns = 0;
  TRYE(hipMemcpyAsync(&ns, d_cnt, 8, hipMemcpyDeviceToHost, s));
  TRYE(hipStreamSynchronize(s));
  e->n_seeds = ns;
  {
    TimerRegion t2(ctx, T_EXT_SORT);
    if ((rc = shn_sort_pairs(ctx, skeys, svals, (uint64_t*)pk2, (uint32_t*)pv2, ns, 0, 2 * t->k))) { shn_ext_destroy(e); return rc; }
    if (ns) {
      hipLaunchKernelGGL(ext_weightkey_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, svals, e->d_weight, ns, skeys);
      if ((rc = shn_sort_pairs(ctx, skeys, svals, (uint64_t*)pk2, (uint32_t*)pv2, ns, 0, 32))) { shn_ext_destroy(e); return rc; }
    }
  }
  TRYE(hipMalloc(&e->d_order, (ns + 1) * 4));
  TRYE(hipMalloc(&e->d_nr, (ns + 1) * 4));
  TRYE(hipMalloc(&e->d_nl, (ns + 1) * 4));
  TRYE(hipMalloc(&e->d_totw, (ns + 1) * 8));
  TRYE(hipMalloc(&e->d_hash, (ns + 1) * 8));
  TRYE(hipMalloc(&e->d_poff, (ns + 1) * 8));
  TRYE(hipMalloc(&e->d_pstored, ns + 1));
  const uint64_t pool_cap = 6 * n + (1ULL << 20);
  const uint64_t fcap = 2 * n + 16;                      // final paths are disjoint: at most one slot per oriented k1-mer
  TRYE(hipMalloc(&e->d_pool, fcap * 4));
  TRYE(hipMemcpyAsync(e->d_order, svals, ns * 4, hipMemcpyDeviceToDevice, s));
  TRYE(hipMemsetAsync(e->d_hash, 0xFF, (ns + 1) * 8, s));
  TRYE(hipMemsetAsync(e->d_nr, 0xFF, (ns + 1) * 4, s));
  TRYE(hipMemsetAsync(e->d_nl, 0, (ns + 1) * 4, s));
  TRYE(hipMemsetAsync(e->d_pstored, 0, ns + 1, s));
  TRYE(hipMemsetAsync(e->d_claim, 0xFF, (2 * n + 1) * 8, s));
  // scratch: second claim buffer, second pool + per-walk plan arrays (double-buffered across iterations)
  void *pprev, *ppool1, *ppool2, *pplan;
  if ((rc = g_shn_ws[24].get((2 * n + 2) * 8, &pprev)) || (rc = g_shn_ws[27].get(pool_cap * 4, &ppool2)) ||
      (rc = g_shn_ws[29].get(pool_cap * 4, &ppool1)) ||
      (rc = g_shn_ws[28].get((ns + 1) * (8 + 8 + 4 + 4 + 1 + 1 + 1) + 64, &pplan))) { shn_ext_destroy(e); return rc; }
  u64 *prev = (u64*)pprev, *cur = e->d_claim2, *fin = e->d_claim;
  uint32_t* pool_a = (uint32_t*)ppool1; uint32_t* pool_b = (uint32_t*)ppool2;
  uint64_t* poff_a = (uint64_t*)pplan; uint64_t* poff_b = poff_a + ns + 1;
  uint32_t* pcap = (uint32_t*)(poff_b + ns + 1);
  uint32_t* long_list = pcap + ns + 1;
  uint8_t* pst_a = (uint8_t*)(long_list + ns + 1); uint8_t* pst_b = pst_a + ns + 1;
  uint8_t* is_long = pst_b + ns + 1;
  TRYE(hipMemsetAsync(prev, 0xFF, (2 * n + 1) * 8, s));
  TRYE(hipMemsetAsync(pst_a, 0, 2 * (ns + 1), s));
  // "prev" memo = (pool_a, poff_a, pst_a); "cur" memo = (pool_b, poff_b, pst_b); swapped every iteration
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  TRYE(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  TRYE(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  TRYE(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  int it = 0;
  uint32_t first = 0;                       // walks [0, first) are final
  bool converged = ns == 0;
  while (!converged && it < max_iterations) {
    TimerRegion t3(ctx, T_EXT_WALK);
    TRYE(hipMemsetAsync(cur, 0xFF, (2 * n + 1) * 8, s));
    TRYE(hipMemsetAsync(d_changed, 0xFF, 4, s));
    TRYE(hipMemsetAsync(d_cnt + 2, 0, 16, s));
    uint64_t nw = ns - first;
    hipLaunchKernelGGL(ext_plan_kernel, dim3((uint32_t)cdiv(nw, 256)), dim3(256), 0, s, e->d_nr, e->d_nl, first, (uint64_t)ns, pool_a,
                       poff_a, pst_a, is_long, long_list, poff_b, pcap, d_cnt + 2, pool_cap);
    unsigned long long plan[2] = {0, 0};
    TRYE(hipMemcpyAsync(plan, d_cnt + 2, 16, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    WalkArgs A;
    A.order = e->d_order; A.adjR = (const Adj4*)e->d_adjR; A.adjL = (const Adj4*)e->d_adjL; A.weight = e->d_weight;
    A.claim_prev = prev; A.claim_cur = cur; A.first = first;
    A.nr_out = e->d_nr; A.nl_out = e->d_nl; A.totw_out = e->d_totw; A.hash_io = e->d_hash; A.changed = d_changed;
    A.pool_prev = pool_a; A.poff_prev = poff_a; A.pstored_prev = pst_a;
    A.pool_cur = pool_b; A.poff_cur = poff_b; A.pcap_cur = pcap; A.pstored_cur = pst_b;
    A.is_long = is_long; A.steps_counter = d_cnt + 1;
    // long (wave per walk) and short (thread per walk) kernels are independent: overlap them on two streams
    if (plan[0]) {
      TRYE(hipEventRecord(ev_fork, s));
      TRYE(hipStreamWaitEvent(aux, ev_fork, 0));
      hipLaunchKernelGGL(ext_walk_long_kernel, dim3((uint32_t)plan[0]), dim3(64), 0, aux, A, long_list);
      TRYE(hipEventRecord(ev_join, aux));
    }
    hipLaunchKernelGGL(ext_walk_kernel<false>, dim3((uint32_t)cdiv(nw, EBLK)), dim3(EBLK), 0, s, A, nw, nullptr, is_long, t->d_keys,
                       t->k, nullptr, nullptr);
    if (plan[0]) TRYE(hipStreamWaitEvent(s, ev_join, 0));
    uint32_t ch = 0;
    TRYE(hipMemcpyAsync(&ch, d_changed, 4, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    it++;
    if (getenv("SHN_DEBUG")) fprintf(stderr, "[shn_extend] iteration %d: first=%u lowest_changed=%u walks=%llu long=%llu\n", it, first, ch,
                                     (unsigned long long)nw, plan[0]);
    // walk q = lowest rank that changed is final now (every lower rank was unchanged, hence final),
    // and so is every walk below it: freeze [first, q] and never recompute them.
    uint32_t newfirst = ch == UNCLAIMED ? (uint32_t)ns : ch + 1;
    hipLaunchKernelGGL(ext_freeze_kernel, dim3((uint32_t)cdiv(2 * n, 256)), dim3(256), 0, s, cur, fin, 2 * n, newfirst);
    if (plan[0]) hipLaunchKernelGGL(ext_keep_final_kernel, dim3((uint32_t)plan[0]), dim3(256), 0, s, long_list, (uint64_t)plan[0], first,
                                    newfirst, pool_b, poff_b, pst_b, e->d_nr, e->d_pool, e->d_poff, e->d_pstored, d_cnt + 4, fcap);
    first = newfirst;
    std::swap(prev, cur);
    std::swap(pool_a, pool_b); std::swap(poff_a, poff_b); std::swap(pst_a, pst_b);
    if (ch == UNCLAIMED) converged = true;
  }
  hipStreamSynchronize(aux);
  hipStreamDestroy(aux);
  hipEventDestroy(ev_fork);
  hipEventDestroy(ev_join);
  e->iterations = it;
  if (!converged) { shn_ext_destroy(e); return shn_fail(SHN_ERR_INTERNAL, "shn_extend: walk fixpoint did not converge"); }
  // e->d_pool / d_poff / d_pstored now hold the paths of every long walk (copied when it became final)
  e->d_claim2 = (cur == (u64*)pprev) ? prev : cur;          // keep the malloc'd scratch (the other one is workspace)
  unsigned long long steps = 0;
  TRYE(hipMemcpyAsync(&steps, d_cnt + 1, 8, hipMemcpyDeviceToHost, s));
  TRYE(hipStreamSynchronize(s));
  e->total_steps = steps;
  TRYE(hipGetLastError());
#undef TRYE
  *out = e;
  return SHN_OK;
}

extern "C" uint64_t shn_ext_n_walks(const shn_ext* e) { return e ? e->n_seeds : 0; }
extern "C" int shn_ext_iterations(const shn_ext* e) { return e ? e->iterations : 0; }
extern "C" uint64_t shn_ext_total_steps(const shn_ext* e) { return e ? e->total_steps : 0; }

extern "C" int shn_ext_stats(shn_ctx* ctx, const shn_ext* e, uint32_t* n_right, uint32_t* n_left, uint64_t* tot_weight) {
  if (!ctx || !e) return shn_fail(SHN_ERR_ARG, "shn_ext_stats: NULL argument");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (n_right) HIP_TRY(hipMemcpyAsync(n_right, e->d_nr, e->n_seeds * 4, hipMemcpyDeviceToHost, s));
  if (n_left) HIP_TRY(hipMemcpyAsync(n_left, e->d_nl, e->n_seeds * 4, hipMemcpyDeviceToHost, s));
  if (tot_weight) HIP_TRY(hipMemcpyAsync(tot_weight, e->d_totw, e->n_seeds * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SHN_OK;
}

extern "C" int shn_ext_emit(shn_ctx* ctx, const shn_ext* e, const uint32_t* ranks, uint64_t n_sel, const uint64_t* offsets,
                            uint8_t* bases_out) {
  if (!ctx || !e || (n_sel && (!ranks || !offsets || !bases_out))) return shn_fail(SHN_ERR_ARG, "shn_ext_emit: NULL argument");
  if (!n_sel) return SHN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  TimerRegion treg(ctx, T_EXTEND);
  uint64_t total = offsets[n_sel];
  uint32_t* d_sel; uint64_t* d_off; uint8_t* d_out;
  HIP_TRY(hipMalloc(&d_sel, n_sel * 4));
  HIP_TRY(hipMalloc(&d_off, (n_sel + 1) * 8));
  HIP_TRY(hipMalloc(&d_out, total + 1));
  HIP_TRY(hipMemcpyAsync(d_sel, ranks, n_sel * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_off, offsets, (n_sel + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(e->d_claim2, 0xFF, (2 * e->n + 1) * 8, s));
  WalkArgs A;
  memset(&A, 0, sizeof(A));
  A.order = e->d_order; A.adjR = (const Adj4*)e->d_adjR; A.adjL = (const Adj4*)e->d_adjL; A.weight = e->d_weight;
  A.claim_prev = e->d_claim; A.claim_cur = e->d_claim2; A.first = 0;
  A.nr_out = e->d_nr; A.nl_out = e->d_nl; A.totw_out = e->d_totw; A.hash_io = e->d_hash;
  // walks with a stored path are expanded in parallel from the pool; the others are re-walked (they are short)
  hipLaunchKernelGGL(ext_emit_stored_kernel, dim3((uint32_t)n_sel), dim3(256), 0, s, d_sel, n_sel, e->d_order, e->d_pstored, e->d_pool,
                     e->d_poff, e->d_nr, e->d_nl, e->table->d_keys, e->k, d_off, d_out);
  hipLaunchKernelGGL(ext_walk_kernel<true>, dim3((uint32_t)cdiv(n_sel, EBLK)), dim3(EBLK), 0, s, A, n_sel, d_sel, e->d_pstored,
                     e->table->d_keys, e->k, d_off, d_out);
  HIP_TRY(hipMemcpyAsync(bases_out, d_out, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  hipFree(d_sel); hipFree(d_off); hipFree(d_out);
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

// weights of arbitrary k1-mer strings in the doubled input (for the `allowed` dict, :404-408)
__global__ void ext_weight_lookup_kernel(const uint64_t* __restrict__ tkeys, const uint64_t* __restrict__ boff, int bits,
                                         const uint32_t* __restrict__ weight, const uint8_t* __restrict__ flags, int k, int canonical,
                                         const uint64_t* __restrict__ q, uint64_t nq, uint32_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  uint64_t key = q[i];
  if (canonical) { uint64_t rc = shn_revcomp(key, k); key = rc < key ? rc : key; }
  int64_t j = shn_table_find(tkeys, boff, bits, key);
  out[i] = (j >= 0 && !(flags[j] & 2)) ? weight[j] : 0;
}

extern "C" int shn_ext_weights(shn_ctx* ctx, const shn_ext* e, const uint64_t* keys, uint64_t n, uint32_t* weights) {
  if (!ctx || !e || (n && (!keys || !weights))) return shn_fail(SHN_ERR_ARG, "shn_ext_weights: NULL argument");
  if (!n) return SHN_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  uint64_t* dq; uint32_t* dw;
  HIP_TRY(hipMalloc(&dq, n * 8));
  HIP_TRY(hipMalloc(&dw, n * 4));
  HIP_TRY(hipMemcpyAsync(dq, keys, n * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ext_weight_lookup_kernel, dim3((uint32_t)cdiv(n, 256)), dim3(256), 0, s, e->table->d_keys,
                     e->table->d_bucket_off, e->table->bits, e->d_weight, e->d_flags, e->k, e->table->canonical, dq, n, dw);
  HIP_TRY(hipMemcpyAsync(weights, dw, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  hipFree(dq); hipFree(dw);
  return SHN_OK;
}
