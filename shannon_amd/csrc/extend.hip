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
// The iteration is change-driven (a worklist): claims persist in one array; after a round every k1-mer
// whose owner changed marks as dirty the walks that examined it -- the old and new owners of the k1-mer and
// of its 8 neighbours, and the walk seeded on it.  Only dirty walks run in the next round, after their old
// claims have been released; the round count stays the dependency depth of the data, but a round costs the
// affected walks, not all of them.  A consistent state (no dirty walk) is the unique fixpoint = the
// sequential result.  Long walks keep their previous path (memo): a wavefront re-checks 64 consecutive old
// steps at once and walks sequentially only from the first changed decision until the path rejoins.
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
  u64* claim;            // live claims: clean walks' + this round's (dirty walks released theirs before the round)
  const u64* claim_old;  // snapshot taken before the round (memo positions of the walk's own old path)
  uint32_t* nr_out; uint32_t* nl_out; uint64_t* totw_out;
  const uint32_t* pool_prev; const uint64_t* poff_prev; const uint8_t* pstored_prev;
  uint32_t* pool_cur; const uint64_t* poff_cur; const uint32_t* pcap_cur; uint8_t* pstored_cur;
  const uint8_t* is_long; const uint8_t* dirty;
  unsigned long long* steps_counter;
};

// One greedy decision (extension_correction.py:223-237): among the candidates that exist and are not
// traversed pick the heaviest, ties in BASES order A,G,C,T (codes 0,2,1,3; strict >).  Traversed = claimed
// by a rank <= r (lower ranks, or this walk's own trail of this round), or -- while re-checking a stretch of
// the walk's old path -- own old position in [lo, hi] (taken from the pre-round snapshot).
__device__ __forceinline__ int decide(const Adj4& cand, uint32_t r, uint32_t lo, uint32_t hi, const u64* claim,
                                      const u64* __restrict__ claim_old, const uint32_t* __restrict__ weight, uint32_t dummy,
                                      uint32_t& bw) {
  u64 cl[4], co[4];
  uint32_t w[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    uint32_t idx = cand.v[b] < 0 ? dummy : (uint32_t)cand.v[b];
    cl[b] = __hip_atomic_load(&claim[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    co[b] = claim_old[idx];
    w[b] = weight[idx >> 1];
  }
  int best = -1;
  bw = 0;
#define CONSIDER(b)                                                                                  \
  if (cand.v[b] >= 0) {                                                                                \
    bool trav = RANK(cl[b]) <= r || RANK(co[b]) < r || (RANK(co[b]) == r && POS(co[b]) >= lo && POS(co[b]) <= hi); \
    if (!trav && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }                                     \
  }
  CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
  return best;
}

// ---- short walks: one thread per walk, one memory round trip per step (candidate rows prefetched).
// EMIT: re-walk the selected final walks against the final claims and write their contig bases.
template <bool EMIT>
__global__ __launch_bounds__(EBLK) void ext_walk_kernel(WalkArgs A, uint64_t n_walks, const uint32_t* __restrict__ list,
                                                        const uint8_t* __restrict__ skip, const u64* __restrict__ final_claim,
                                                        const uint64_t* __restrict__ tkeys, int k,
                                                        const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out_bases) {
  __shared__ unsigned long long blk_steps;
  if (threadIdx.x == 0) blk_steps = 0;
  __syncthreads();
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mysteps = 0;
  if (t < n_walks) {
    const uint32_t r = list[t];
    if (!(skip && skip[r])) {
      const uint32_t o = A.order[r];
      uint32_t nr = 0, nl = 0;
      uint64_t tot = 0;
      // final_claim: the pre-round snapshot (or, for EMIT, the converged claims); A.claim: live claims of this
      // round (EMIT: a scratch array for the own trail)
      bool isvoid = RANK(final_claim[o]) < r || (!EMIT && RANK(__hip_atomic_load(&A.claim[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r);
      uint8_t* dst = nullptr;
      uint32_t nl_known = 0, nr_known = 0;
      if (EMIT && !isvoid) {
        dst = out_bases + out_off[t];
        nl_known = A.nl_out[r];
        nr_known = A.nr_out[r];
        uint64_t s = oriented_string(tkeys, o, k);
        for (int j = 0; j < k; j++) dst[nl_known + j] = "ACGT"[(s >> (2 * (k - 1 - j))) & 3];
      }
      if (!isvoid) {
        atomicMin(&A.claim[o], CLAIM(r, 0));
        tot = A.weight[o >> 1];
        uint32_t pos = 0;
        for (int dir = 0; dir < 2; dir++) {
          const Adj4* adj = dir == 0 ? A.adjR : A.adjL;
          uint32_t steps = 0;
          Adj4 cand = adj[o];
          while (true) {
            u64 cl[4], cf[4];
            uint32_t w[4];
            Adj4 nxt[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
              uint32_t idx = cand.v[b] < 0 ? o : (uint32_t)cand.v[b];
              cl[b] = __hip_atomic_load(&A.claim[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              cf[b] = final_claim[idx];
              w[b] = A.weight[idx >> 1];
              nxt[b] = adj[idx];
            }
            int best = -1;
            uint32_t bw = 0;
#define CONSIDER(b) if (cand.v[b] >= 0 && RANK(cl[b]) > r && RANK(cf[b]) >= r && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }
            CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
            if (best < 0) break;
            uint32_t nbest = (uint32_t)cand.v[best];
            pos++;
            __hip_atomic_fetch_min(&A.claim[nbest], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (EMIT) {
              const uint32_t lim = dir == 0 ? nr_known : nl_known;
              if (steps >= lim) { atomicAdd(A.steps_counter, 1ULL); break; }      // re-walk left its recorded path: flag, do not write
              if (dir == 0) dst[nl_known + k + steps] = "ACGT"[best];
              else dst[nl_known - 1 - steps] = "ACGT"[best];
            }
            steps++;
            tot += bw;
            cand = best == 0 ? nxt[0] : best == 1 ? nxt[1] : best == 2 ? nxt[2] : nxt[3];
          }
          if (EMIT && steps != (dir == 0 ? nr_known : nl_known)) atomicAdd(A.steps_counter, 1ULL << 32);
          if (dir == 0) nr = steps; else nl = steps;
        }
      }
      if (!EMIT) {
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

// ---- long walks: one wavefront per walk.  Clean walks only carry their memo forward; dirty walks re-check
// the old path 64 steps at a time and walk sequentially from the first changed decision until they rejoin it.
#define NONE32 0xFFFFFFFFu
__global__ __launch_bounds__(64) void ext_walk_long_kernel(WalkArgs A, const uint32_t* __restrict__ long_list) {
  const uint32_t r = long_list[blockIdx.x];
  const int lane = threadIdx.x;
  const uint32_t o = A.order[r];
  // memo layout in the pool: [nR, nL, node of step 0, node of step 1, ...]
  const bool had = A.pstored_prev[r] != 0;
  const uint32_t* oldhdr = A.pool_prev + (had ? A.poff_prev[r] : 0);
  const uint32_t* oldp = oldhdr + 2;
  const uint32_t oldR = had ? oldhdr[0] : 0, oldL = had ? oldhdr[1] : 0;
  uint32_t* newhdr = A.pool_cur + A.poff_cur[r];
  uint32_t* newp = newhdr + 2;
  const bool room = A.pcap_cur[r] >= 2;
  const uint32_t cap = room ? A.pcap_cur[r] - 2 : 0;
  const bool run = A.dirty[r] != 0;
  const bool isvoid = run && (RANK(A.claim_old[o]) < r || RANK(__hip_atomic_load(&A.claim[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r);
  if (!run || isvoid) {
    // clean (results unchanged) or seed currently traversed: keep the memo, the walk may need it later
    bool keep = false;
    if (had && room && oldR + oldL <= cap) {
      for (uint32_t i = lane; i < oldR + oldL; i += 64) newp[i] = oldp[i];
      if (lane == 0) { newhdr[0] = oldR; newhdr[1] = oldL; }
      keep = true;
    }
    if (lane == 0) {
      A.pstored_cur[r] = keep ? 1 : 0;
      if (isvoid) { A.nr_out[r] = UNCLAIMED; A.nl_out[r] = 0; A.totw_out[r] = 0; }
    }
    return;
  }
  uint32_t ns = 0, nr_new = 0;
  uint64_t tot = A.weight[o >> 1];
  if (lane == 0) atomicMin(&A.claim[o], CLAIM(r, 0));
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
        const bool is_term = (uint32_t)lane == nchunk && nchunk < 64u;   // one past the old end: must decide "stop"
        const bool checked = (uint32_t)lane < nchunk || is_term;
        int32_t chosen = -1;
        uint32_t bw = 0;
        if (checked) {
          uint32_t before = lane == 0 ? cur : oldp[s - 1];
          Adj4 cand = adj[before];
          int b = decide(cand, r, oi + 1, s, A.claim, A.claim_old, A.weight, o, bw);
          chosen = b < 0 ? -1 : cand.v[b];
        }
        const int32_t expect = (checked && !is_term) ? (int32_t)oldp[s] : -1;
        const u64 bad = __ballot(checked && chosen != expect);
        const uint32_t m = bad ? (uint32_t)(__ffsll((long long)bad) - 1) : 64u;
        const uint32_t conf = min(m, nchunk);               // confirmed old steps: lanes [0, conf)
        uint64_t myw = 0;
        if ((uint32_t)lane < conf) {
          uint32_t node = oldp[s];
          uint32_t pos = ns + lane + 1;
          __hip_atomic_fetch_min(&A.claim[node], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ns + lane < cap) newp[ns + lane] = node;
          myw = A.weight[node >> 1];
        }
        for (int off = 32; off > 0; off >>= 1) myw += __shfl_xor(myw, off, 64);
        tot += myw;
        if (conf > 0) cur = oldp[oi + conf - 1];
        ns += conf;
        oi += conf;
        if (m == 64u) {
          if (nchunk < 64u) break;            // terminal lane agreed: the old path is still complete
          continue;
        }
        taken = __shfl(chosen, (int)m, 64);
        taken_w = __shfl(bw, (int)m, 64);
        if (taken < 0) break;                 // the walk now stops here
      } else {
        Adj4 cand = adj[cur];
        uint32_t bw = 0;
        int b = decide(cand, r, 1, 0, A.claim, A.claim_old, A.weight, o, bw);
        if (b < 0) break;
        taken = cand.v[b];
        taken_w = bw;
      }
      // take `taken` as the next step, then look for a rejoin with the old path
      {
        uint32_t pos = ns + 1;
        if (lane == 0) {
          __hip_atomic_fetch_min(&A.claim[taken], CLAIM(r, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ns < cap) newp[ns] = (uint32_t)taken;
        }
        tot += taken_w;
        ns++;
        cur = (uint32_t)taken;
        oi = NONE32;
        if (had) {
          u64 cpv = A.claim_old[taken];
          if (RANK(cpv) == r && POS(cpv) >= 1) {
            uint32_t p = POS(cpv) - 1;
            if (p >= ob && p < oe && oldp[p] == (uint32_t)taken) oi = p + 1;
          }
        }
      }
    }
    if (dir == 0) nr_new = ns;
  }
  const bool keep = room && ns <= cap;
  if (lane == 0) {
    if (keep) { newhdr[0] = nr_new; newhdr[1] = ns - nr_new; }
    A.nr_out[r] = nr_new;
    A.nl_out[r] = ns - nr_new;
    A.totw_out[r] = tot;
    A.pstored_cur[r] = keep ? 1 : 0;
    if (ns) atomicAdd(A.steps_counter, (unsigned long long)ns);
  }
}

// classify the walks for the next round and lay out the path pool: long walks (memo or length) go to the
// wavefront kernel (dirty or not -- clean ones only copy their memo); dirty short walks go to the thread kernel
__global__ void ext_plan_kernel(const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, uint64_t ns, uint32_t frozen,
                                const uint32_t* __restrict__ pool_prev, const uint64_t* __restrict__ poff_prev,
                                const uint8_t* __restrict__ pstored_prev, const uint8_t* __restrict__ dirty,
                                uint8_t* __restrict__ is_long, uint32_t* __restrict__ long_list, uint32_t* __restrict__ short_list,
                                uint64_t* __restrict__ poff, uint32_t* __restrict__ pcap, unsigned long long* __restrict__ counters,
                                uint64_t pool_cap) {
  // ns here = current rank limit (walks >= limit have not started yet); walks < frozen are final and never run
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + frozen;
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
  } else if (dirty[r]) {
    unsigned long long idx = atomicAdd(&counters[2], 1ULL);
    short_list[idx] = (uint32_t)r;
  }
  is_long[r] = lg;
}

// drop the claims of the walks that are about to re-run
__global__ void ext_release_kernel(u64* __restrict__ claim, uint64_t n2, const uint8_t* __restrict__ dirty, uint64_t ns) {
  uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n2) return;
  uint32_t rk = RANK(claim[o]);
  if (rk != UNCLAIMED && rk < ns && dirty[rk]) claim[o] = UNCLAIMED64;
}

// after a round: every k1-mer whose owner changed dirties the walks that looked at it
__global__ void ext_mark_kernel(const u64* __restrict__ claim, const u64* __restrict__ claim_old, uint64_t n2,
                                const Adj4* __restrict__ adjR, const Adj4* __restrict__ adjL, const uint32_t* __restrict__ seed_rank,
                                uint8_t* __restrict__ dirty, const uint8_t* __restrict__ ran, uint32_t* __restrict__ owned,
                                unsigned long long* __restrict__ n_changed, uint32_t frozen, uint32_t limit) {
  uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a = UNCLAIMED, b = UNCLAIMED;
  if (y < n2) { a = RANK(claim_old[y]); b = RANK(claim[y]); }
  // one atomic per wavefront for the change counter (it was one per changed k1-mer on a single address)
  unsigned long long chm = __ballot(a != b);
  if (chm && (threadIdx.x & 63) == (uint32_t)(__ffsll((long long)chm) - 1)) atomicAdd(n_changed, (unsigned long long)__popcll(chm));
  if (y >= n2) return;
  if (b != UNCLAIMED && ran[b]) atomicAdd(&owned[b], 1u);
  if (a == b) return;
  // a walk depends only on lower ranks: walks below `frozen` are final whatever happens above them; walks at or
  // above `limit` have not started (they all run when their phase opens)
#define MARK(x) if ((x) >= frozen && (x) < limit) dirty[x] = 1
  MARK(a);
  MARK(b);
  uint32_t sr = seed_rank[y];
  MARK(sr);
  Adj4 L = adjL[y], R = adjR[y];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    int32_t nb = q < 4 ? L.v[q] : R.v[q - 4];
    if (nb < 0) continue;
    uint32_t x = RANK(claim_old[nb]), z = RANK(claim[nb]);
    MARK(x);
    MARK(z);
  }
#undef MARK
}

// A walk that ran this round must own exactly the k1-mers on the path it recorded; if a lower rank took one
// of them back during the round (the walk saw it free for a moment) its record is stale: run it again.
__global__ void ext_verify_kernel(const uint8_t* __restrict__ ran, const uint32_t* __restrict__ owned, const uint32_t* __restrict__ nr,
                                  const uint32_t* __restrict__ nl, uint64_t ns, uint8_t* __restrict__ dirty,
                                  unsigned long long* __restrict__ n_unstable) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ns || !ran[r]) return;
  uint32_t expect = nr[r] == UNCLAIMED ? 0u : nr[r] + nl[r] + 1u;
  if (owned[r] != expect) { dirty[r] = 1; atomicAdd(n_unstable, 1ULL); }
}

__global__ void ext_seed_rank_kernel(const uint32_t* __restrict__ order, uint64_t ns, uint32_t* __restrict__ seed_rank) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < ns) seed_rank[order[r]] = (uint32_t)r;
}

// copy the memo of every stored long walk into the ext-owned final pool (after convergence).  A memo is used
// only if it is exactly the final path: header == final lengths and every node carries the claim (r, step+1).
__global__ void ext_keep_final_kernel(const uint32_t* __restrict__ long_list, uint64_t n_long, const uint32_t* __restrict__ pool,
                                      const uint64_t* __restrict__ poff, const uint8_t* __restrict__ pstored,
                                      const uint32_t* __restrict__ nr, const uint32_t* __restrict__ nl, const u64* __restrict__ claim,
                                      uint32_t* __restrict__ fpool, uint64_t* __restrict__ foff, uint8_t* __restrict__ fstored,
                                      unsigned long long* __restrict__ cursor, uint64_t fcap, unsigned long long* __restrict__ n_bad) {
  if (blockIdx.x >= n_long) return;
  uint32_t r = long_list[blockIdx.x];
  if (!pstored[r] || nr[r] == UNCLAIMED) return;
  const uint32_t* h = pool + poff[r];
  uint32_t len = h[0] + h[1];
  __shared__ unsigned long long base;
  __shared__ int ok;
  if (threadIdx.x == 0) ok = (h[0] == nr[r] && h[1] == nl[r]) ? 1 : 0;
  __syncthreads();
  if (ok) for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) if (claim[h[2 + i]] != CLAIM(r, i + 1)) ok = 0;
  __syncthreads();
  if (!ok) { if (threadIdx.x == 0) atomicAdd(n_bad, 1ULL); return; }
  if (threadIdx.x == 0) base = atomicAdd(cursor, (unsigned long long)len);
  __syncthreads();
  if (base + len > fcap) return;
  for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) fpool[base + i] = h[2 + i];
  if (threadIdx.x == 0) { foff[r] = base; fstored[r] = 1; }
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
  unsigned long long* d_cnt = (unsigned long long*)pc;      // [0] seeds [1] steps [2..4] plan (long, pool, short) [5] final pool [6] changed
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
  TRYE(hipMalloc(&e->d_poff, (ns + 1) * 8));
  TRYE(hipMalloc(&e->d_pstored, ns + 1));
  const uint64_t pool_cap = 6 * n + (1ULL << 20);
  const uint64_t fcap = 2 * n + 16;                      // final paths are disjoint: at most one slot per oriented k1-mer
  TRYE(hipMalloc(&e->d_pool, fcap * 4));
  TRYE(hipMemcpyAsync(e->d_order, svals, ns * 4, hipMemcpyDeviceToDevice, s));
  TRYE(hipMemsetAsync(e->d_nr, 0xFF, (ns + 1) * 4, s));
  TRYE(hipMemsetAsync(e->d_nl, 0, (ns + 1) * 4, s));
  TRYE(hipMemsetAsync(e->d_totw, 0, (ns + 1) * 8, s));
  TRYE(hipMemsetAsync(e->d_pstored, 0, ns + 1, s));
  TRYE(hipMemsetAsync(e->d_claim, 0xFF, (2 * n + 1) * 8, s));
  // scratch: claim snapshot, two memo pools + per-walk plan arrays (double-buffered across rounds)
  void *ppool1, *ppool2, *pplan, *pseed;
  if ((rc = g_shn_ws[27].get(pool_cap * 4, &ppool2)) || (rc = g_shn_ws[29].get(pool_cap * 4, &ppool1)) ||
      (rc = g_shn_ws[28].get((ns + 1) * (8 + 8 + 4 + 4 + 4 + 4 + 1 + 1 + 1 + 1 + 1) + 64, &pplan)) ||
      (rc = g_shn_ws[24].get((2 * n + 2) * 4, &pseed))) { shn_ext_destroy(e); return rc; }
  u64 *claim = e->d_claim, *snap = e->d_claim2;
  uint32_t* pool_a = (uint32_t*)ppool1; uint32_t* pool_b = (uint32_t*)ppool2;
  uint64_t* poff_a = (uint64_t*)pplan; uint64_t* poff_b = poff_a + ns + 1;
  uint32_t* pcap = (uint32_t*)(poff_b + ns + 1);
  uint32_t* long_list = pcap + ns + 1;
  uint32_t* short_list = long_list + ns + 1;
  uint32_t* owned = short_list + ns + 1;
  uint8_t* pst_a = (uint8_t*)(owned + ns + 1); uint8_t* pst_b = pst_a + ns + 1;
  uint8_t* is_long = pst_b + ns + 1;
  uint8_t* dirty = is_long + ns + 1;
  uint8_t* ran = dirty + ns + 1;
  uint32_t* seed_rank = (uint32_t*)pseed;
  TRYE(hipMemsetAsync(pst_a, 0, 2 * (ns + 1), s));
  TRYE(hipMemsetAsync(seed_rank, 0xFF, (2 * n + 1) * 4, s));
  if (ns) hipLaunchKernelGGL(ext_seed_rank_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, e->d_order, (uint64_t)ns, seed_rank);
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  TRYE(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  TRYE(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  TRYE(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  int it = 0;
  bool converged = ns == 0;
  unsigned long long last_long = 0;
  const uint32_t g2n = (uint32_t)cdiv(2 * n, 256);
  // Rank phases: a walk depends only on lower ranks, so the fixpoint is reached block by block -- first the
  // heaviest seeds (where the long, mutually dependent walks live), then geometrically larger blocks that see
  // final lower ranks and settle in a few rounds.
  uint32_t frozen = 0, limit = (uint32_t)std::min<unsigned long long>(ns, std::max<unsigned long long>(ns / 32, 4096));
  TRYE(hipMemsetAsync(dirty, 0, ns + 1, s));
  TRYE(hipMemsetAsync(dirty, 1, limit, s));
  while (!converged && it < max_iterations) {
    TimerRegion t3(ctx, T_EXT_WALK);
    // snapshot, then release the claims of the walks that re-run this round
    TRYE(hipMemcpyAsync(snap, claim, (2 * n + 1) * 8, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(ext_release_kernel, dim3(g2n), dim3(256), 0, s, claim, 2 * n, dirty, (uint64_t)ns);
    TRYE(hipMemsetAsync(d_cnt + 2, 0, 24, s));
    TRYE(hipMemsetAsync(d_cnt + 6, 0, 16, s));
    if (limit > frozen)
      hipLaunchKernelGGL(ext_plan_kernel, dim3((uint32_t)cdiv(limit - frozen, 256)), dim3(256), 0, s, e->d_nr, e->d_nl, (uint64_t)limit, frozen,
                         pool_a, poff_a, pst_a, dirty, is_long, long_list, short_list, poff_b, pcap, d_cnt + 2, pool_cap);
    unsigned long long plan[3] = {0, 0, 0};
    TRYE(hipMemcpyAsync(plan, d_cnt + 2, 24, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    WalkArgs A;
    A.order = e->d_order; A.adjR = (const Adj4*)e->d_adjR; A.adjL = (const Adj4*)e->d_adjL; A.weight = e->d_weight;
    A.claim = claim; A.claim_old = snap;
    A.nr_out = e->d_nr; A.nl_out = e->d_nl; A.totw_out = e->d_totw;
    A.pool_prev = pool_a; A.poff_prev = poff_a; A.pstored_prev = pst_a;
    A.pool_cur = pool_b; A.poff_cur = poff_b; A.pcap_cur = pcap; A.pstored_cur = pst_b;
    A.is_long = is_long; A.dirty = dirty; A.steps_counter = d_cnt + 1;
    // long (wave per walk) and short (thread per walk) kernels are independent: overlap them on two streams
    if (plan[0]) {
      TRYE(hipEventRecord(ev_fork, s));
      TRYE(hipStreamWaitEvent(aux, ev_fork, 0));
      hipLaunchKernelGGL(ext_walk_long_kernel, dim3((uint32_t)plan[0]), dim3(64), 0, aux, A, long_list);
      TRYE(hipEventRecord(ev_join, aux));
    }
    if (plan[2]) hipLaunchKernelGGL(ext_walk_kernel<false>, dim3((uint32_t)cdiv(plan[2], EBLK)), dim3(EBLK), 0, s, A, (uint64_t)plan[2],
                                    short_list, nullptr, snap, t->d_keys, t->k, nullptr, nullptr);
    if (plan[0]) TRYE(hipStreamWaitEvent(s, ev_join, 0));
    // who has to run next round?
    TRYE(hipMemcpyAsync(ran, dirty, ns + 1, hipMemcpyDeviceToDevice, s));
    TRYE(hipMemsetAsync(dirty, 0, ns + 1, s));
    TRYE(hipMemsetAsync(owned, 0, (ns + 1) * 4, s));
    hipLaunchKernelGGL(ext_mark_kernel, dim3(g2n), dim3(256), 0, s, claim, snap, 2 * n, (const Adj4*)e->d_adjR, (const Adj4*)e->d_adjL,
                       seed_rank, dirty, ran, owned, d_cnt + 6, frozen, limit);
    hipLaunchKernelGGL(ext_verify_kernel, dim3((uint32_t)cdiv(ns, 256)), dim3(256), 0, s, ran, owned, e->d_nr, e->d_nl, (uint64_t)ns, dirty,
                       d_cnt + 7);
    unsigned long long chg[2] = {0, 0};
    TRYE(hipMemcpyAsync(chg, d_cnt + 6, 16, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    unsigned long long nchanged = chg[0] + chg[1];
    if (getenv("SHN_EXT_ALLDIRTY")) { TRYE(hipMemsetAsync(dirty, 0, ns + 1, s)); TRYE(hipMemsetAsync(dirty + frozen, 1, limit - frozen, s)); }
    it++;
    last_long = plan[0];
    if (getenv("SHN_DEBUG")) fprintf(stderr, "[shn_extend] round %d [%u,%u): long=%llu short_dirty=%llu changed_kmers=%llu unstable=%llu\n", it, frozen, limit,
                                     plan[0], plan[2], chg[0], chg[1]);
    std::swap(pool_a, pool_b); std::swap(poff_a, poff_b); std::swap(pst_a, pst_b);
    if (nchanged == 0) {
      if (limit >= ns) converged = true;
      else {                                   // this block is final: open the next one
        frozen = limit;
        limit = (uint32_t)std::min<unsigned long long>(ns, (unsigned long long)limit * 4);
        TRYE(hipMemsetAsync(dirty + frozen, 1, limit - frozen, s));
      }
    }
  }
  hipStreamSynchronize(aux);
  hipStreamDestroy(aux);
  hipEventDestroy(ev_fork);
  hipEventDestroy(ev_join);
  e->iterations = it;
  if (!converged) { shn_ext_destroy(e); return shn_fail(SHN_ERR_INTERNAL, "shn_extend: walk fixpoint did not converge"); }
  // keep the paths of the long walks for the emit (memo of the last round = pool_a after the swap)
  if (last_long) hipLaunchKernelGGL(ext_keep_final_kernel, dim3((uint32_t)last_long), dim3(256), 0, s, long_list, (uint64_t)last_long, pool_a,
                                    poff_a, pst_a, e->d_nr, e->d_nl, claim, e->d_pool, e->d_poff, e->d_pstored, d_cnt + 5, fcap, d_cnt + 8);
  unsigned long long steps = 0, nbad = 0;
  TRYE(hipMemcpyAsync(&steps, d_cnt + 1, 8, hipMemcpyDeviceToHost, s));
  TRYE(hipMemcpyAsync(&nbad, d_cnt + 8, 8, hipMemcpyDeviceToHost, s));
  TRYE(hipStreamSynchronize(s));
  if (getenv("SHN_DEBUG")) fprintf(stderr, "[shn_extend] converged after %d rounds; %llu long walks, %llu memos not final (re-walked at emit)\n", it, last_long, nbad);
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
  A.claim = e->d_claim2;                      // scratch: own trail of the re-walk
  A.nr_out = e->d_nr; A.nl_out = e->d_nl; A.totw_out = e->d_totw;
  unsigned long long* d_flag = nullptr;
  HIP_TRY(hipMalloc(&d_flag, 8));
  HIP_TRY(hipMemsetAsync(d_flag, 0, 8, s));
  A.steps_counter = d_flag;                  // EMIT: counts re-walks that disagree with the recorded path
  // walks with a stored path are expanded in parallel from the pool; the others are re-walked (they are short)
  // Default: re-walk every selected walk against the final claims (verified deterministic).  The parallel
  // expansion from the stored paths is experimental (SHN_EMIT_STORED=1): batched launches showed a 64-entry
  // stale chunk in a few contigs that is not understood yet.
  const bool seq_only = getenv("SHN_EMIT_STORED") == nullptr;
  if (!seq_only)
    hipLaunchKernelGGL(ext_emit_stored_kernel, dim3((uint32_t)n_sel), dim3(256), 0, s, d_sel, n_sel, e->d_order, e->d_pstored, e->d_pool,
                       e->d_poff, e->d_nr, e->d_nl, e->table->d_keys, e->k, d_off, d_out);
  hipLaunchKernelGGL(ext_walk_kernel<true>, dim3((uint32_t)cdiv(n_sel, EBLK)), dim3(EBLK), 0, s, A, n_sel, d_sel,
                     seq_only ? nullptr : e->d_pstored, e->d_claim, e->table->d_keys, e->k, d_off, d_out);
  HIP_TRY(hipMemcpyAsync(bases_out, d_out, total, hipMemcpyDeviceToHost, s));
  unsigned long long flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, d_flag, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  hipFree(d_sel); hipFree(d_off); hipFree(d_out); hipFree(d_flag);
  if (flag) return shn_fail(SHN_ERR_INTERNAL, "shn_ext_emit: a re-walked contig disagrees with its recorded path (overlong=" +
                            std::to_string(flag & 0xFFFFFFFFULL) + ", length mismatches=" + std::to_string(flag >> 32) + ")");
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
