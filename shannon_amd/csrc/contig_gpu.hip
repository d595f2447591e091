// The contig stage of extension_correction.run_correction (rows a5-a6) for inputs with hundreds of thousands of
// candidate contigs: duplicate_check (extension_correction.py:247-270) and contig_connections (:372-397), same results
// as the sequential stage of contig_host.hip (shn_contig_graph), with the bulk of the work on the GPU.
//
// duplicate_check: whether candidate c is a duplicate depends only on the ACCEPTED earlier candidates -- their r-mer hit counts
// in c, the one with the most hits (ties: the one whose last hit comes last, `>=` at :258) and how much of c its shared r-mers
// cover.  The sequential result is the only assignment acc[] with acc[c] = decide(c | acc[< c]).  Here the r-mers of ALL
// candidates are sorted once on the GPU; candidates are taken in seed-order blocks, and inside a block the decisions are iterated
// (all candidates of the block at once, against the frozen decisions of the earlier blocks and the tentative ones of the block)
// until none changes -- the fixpoint of a block on top of a final prefix is the sequential result.  Per block: the index of the
// block = the sorted entries of the accepted earlier candidates and of all the block's own (a look-up skips the block's candidates
// that are not accepted at the moment).  Per round: every window of the block walks the entries of its run before it and adds
// (candidate, parent) hits to a pair table (count, last hit position), the best parent of a candidate is a 64-bit atomicMax over
// its pairs, a second pass marks the windows that hit it, coverage is summed per candidate.  After a block's first round only the
// candidates behind a changed decision in one of their runs are evaluated again.  Variants of a highly expressed transcript
// come long after it in the seed order (their weight is the error rate times its weight), so they meet it frozen and fall in
// their block's first round.
// contig_connections: the K-mers of the accepted contigs are sorted on the GPU; only K-mers occurring in two different
// contigs matter, those runs are compacted and sent to the host, which replays the reference's loop over them
// (neighbour lists in dict insertion order, weights counted per position pair).
#include "contig_graph.h"
#include <atomic>
#include <mutex>

namespace {

#define CG_BLK 256

// contig id of every base: cid[g] for off[c] <= g < off[c+1]
__global__ void cg_cid_kernel(const uint64_t* __restrict__ off, uint64_t n_cand, uint32_t* __restrict__ cid) {
  // one wavefront per contig: contigs are 75 .. a few thousand bases
  const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (c >= n_cand) return;
  const uint64_t lo = off[c], hi = off[c + 1];
  for (uint64_t g = lo + (threadIdx.x & 63); g < hi; g += 64) cid[g] = (uint32_t)c;
}

// flag[g] = 1 iff a k-window starts at base g inside its contig, the contig is selected (use == NULL: all) and the
// window holds ACGT only
__global__ void cg_flag_kernel(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, const uint32_t* __restrict__ cid,
                               const int32_t* __restrict__ use, uint64_t total, int k, uint32_t* __restrict__ flag) {
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = cid[g];
    flag[g] = (g + (uint64_t)k <= off[c + 1] && (!use || use[c] != 0)) ? 1u : 0u;
  }
}

__device__ __forceinline__ uint32_t cg_code(uint8_t b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; }

__global__ void cg_keys_kernel(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ pos,
                               uint64_t total, int k, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                               unsigned long long* __restrict__ n_bad) {
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    if (!flag[g]) continue;
    uint64_t key = 0;
    bool bad = false;
    for (int j = 0; j < k; j++) { const uint32_t c = cg_code(bases[g + j]); bad |= c > 3; key = (key << 2) | (uint64_t)(c & 3); }
    if (bad) atomicAdd(n_bad, 1ULL);
    keys[pos[g]] = key;
    vals[pos[g]] = (uint32_t)g;
  }
}

// k <= 16: the window's key above its base index in one word -- the sort then moves 8 bytes per entry and pass, not 12
__global__ void cg_keys_packed_kernel(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ pos,
                                      uint64_t total, int k, uint64_t* __restrict__ words, unsigned long long* __restrict__ n_bad) {
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    if (!flag[g]) continue;
    uint64_t key = 0;
    bool bad = false;
    for (int j = 0; j < k; j++) { const uint32_t c = cg_code(bases[g + j]); bad |= c > 3; key = (key << 2) | (uint64_t)(c & 3); }
    if (bad) atomicAdd(n_bad, 1ULL);
    words[pos[g]] = (key << 32) | (uint64_t)(uint32_t)g;
  }
}

// candidate id + accepted flag of every sorted entry
__global__ void cg_scid_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ cid, uint64_t n, uint32_t* __restrict__ scid) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) scid[i] = cid[vals[i]];
}
// the index of a block [lo, hi): the entries of the accepted candidates before it (final) and ALL entries of the block's own
// candidates (tentative: a look-up skips those whose candidate is not accepted at the moment) -- built once per block
__global__ void cg_accflag_kernel(const uint32_t* __restrict__ scid, const uint8_t* __restrict__ acc, uint64_t n, uint32_t lo, uint32_t hi,
                                  uint32_t* __restrict__ flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = scid[i];
    flag[i] = c < lo ? acc[c] : c < hi ? 1u : 0u;
  }
}
// rounds after a block's first: only the candidates that share an r-mer with an earlier candidate of the block whose decision
// changed in the last round can come out differently (decide(c) reads the decisions of the candidates before c in c's runs and
// nothing else) -- the entries of the changed candidates flag the candidates behind them in their runs
__global__ void cg_affected_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ acand, uint64_t na, uint32_t lo,
                                   const uint8_t* __restrict__ chg, uint8_t* __restrict__ aff) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t d = acand[i];
    if (d < lo || !chg[d]) continue;
    const uint64_t key = akey[i];
    for (uint64_t j = i + 1; j < na && akey[j] == key; j++) {
      const uint32_t c = acand[j];
      if (c != d) aff[c] = 1;
    }
  }
}
// the per-candidate state of the candidates evaluated this round back to zero
__global__ void cg_clear_kernel(const uint8_t* __restrict__ aff, const uint32_t* __restrict__ cid, uint64_t g_lo, uint64_t g_hi, uint32_t lo, uint32_t hi,
                                uint8_t* __restrict__ hit, unsigned long long* __restrict__ best, uint32_t* __restrict__ cov) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t g = g_lo + t; g < g_hi; g += stride) if (aff[cid[g]]) hit[g] = 0;
  for (uint64_t c = lo + t; c < hi; c += stride) if (aff[c]) { best[c] = 0; cov[c] = 0; }
}
__global__ void cg_acc_compact_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ scid, const uint32_t* __restrict__ flag,
                                      const uint32_t* __restrict__ vals, const uint64_t* __restrict__ apos, uint64_t n, uint64_t* __restrict__ akey,
                                      uint32_t* __restrict__ acand, uint32_t* __restrict__ aval) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (flag[i]) { akey[apos[i]] = keys[i]; acand[apos[i]] = scid[i]; aval[apos[i]] = vals[i]; }
}

// (candidate, parent) -> hit count and position of the last hit.  Open addressing; key = (c + 1) << 32 | p, 0 = empty.
struct PairSlot { unsigned long long key; uint32_t count; uint32_t last; };
__device__ __forceinline__ bool pair_add(PairSlot* __restrict__ tab, uint64_t mask, uint32_t c, uint32_t p, uint32_t pos) {
  const unsigned long long kk = ((unsigned long long)(c + 1) << 32) | p;
  uint64_t s = shn_mix64(kk) & mask;
  for (int probe = 0; probe < 1024; probe++) {
    unsigned long long cur = __hip_atomic_load(&tab[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) { unsigned long long old = atomicCAS(&tab[s].key, 0ULL, kk); cur = old == 0 ? kk : old; }
    if (cur == kk) { atomicAdd(&tab[s].count, 1u); atomicMax(&tab[s].last, pos); return true; }
    s = (s + 1) & mask;
  }
  return false;
}
// every window of the open block (an index entry of a candidate >= lo): the accepted entries of its run that come before it are its
// hits (:250-259)
__global__ void cg_hits_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ acand, const uint32_t* __restrict__ aval, uint64_t na,
                               const uint64_t* __restrict__ off, uint32_t lo, const uint8_t* __restrict__ acc, const uint8_t* __restrict__ aff,
                               PairSlot* __restrict__ tab, uint64_t mask, uint32_t* __restrict__ overflow) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = acand[i];
    if (c < lo || !aff[c]) continue;
    const uint64_t key = akey[i];
    const uint32_t pos = (uint32_t)(aval[i] - off[c]);
    for (uint64_t j = i; j-- > 0 && akey[j] == key;) {
      const uint32_t p = acand[j];
      if (p == c) continue;                       // the candidate's own earlier windows (it is not in the index while it is checked)
      if (p >= lo && !acc[p]) continue;           // a candidate of the block that is not accepted at the moment
      if (!pair_add(tab, mask, c, p, pos)) atomicOr(overflow, 1u);
    }
  }
}
// best parent of every candidate: most hits; among equals the one whose last hit comes last (`>=`: the latest wins, :258-259),
// two parents hitting in the same window are listed in acceptance order, so the later candidate wins.
// pack: count (22 bits) | last hit position (21) | parent (21)
__global__ void cg_best_kernel(const PairSlot* __restrict__ tab, uint64_t slots, unsigned long long* __restrict__ best) {
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = tab[s].key;
    if (!k) continue;
    const uint32_t c = (uint32_t)(k >> 32) - 1, p = (uint32_t)k;
    const uint32_t cnt = tab[s].count > 0x3FFFFFu ? 0x3FFFFFu : tab[s].count;
    atomicMax(&best[c], ((unsigned long long)cnt << 42) | ((unsigned long long)tab[s].last << 21) | (unsigned long long)p);
  }
}
// windows of the open block whose r-mer occurs in the candidate's best parent (:262-265)
__global__ void cg_cover_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ acand, const uint32_t* __restrict__ aval, uint64_t na,
                                uint32_t lo, const unsigned long long* __restrict__ best, const uint8_t* __restrict__ aff, uint8_t* __restrict__ hit) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = acand[i];
    if (c < lo || !aff[c]) continue;
    const unsigned long long b = best[c];
    if (!b) continue;
    const uint32_t bp = (uint32_t)(b & 0x1FFFFFu);
    const uint64_t key = akey[i];
    for (uint64_t j = i; j-- > 0 && akey[j] == key;)
      if (acand[j] == bp) { hit[aval[i]] = 1; break; }
  }
}
// covered bases of a candidate = bases under at least one hit window (a[i:i+r] = 1, :264-265)
__global__ void cg_covsum_kernel(const uint8_t* __restrict__ hit, const uint32_t* __restrict__ cid, const uint64_t* __restrict__ off,
                                 uint64_t g_lo, uint64_t g_hi, int r, const uint8_t* __restrict__ aff, uint32_t* __restrict__ cov) {
  // (every lane of a wavefront takes part in every trip: the 64 bases of a trip mostly belong to one candidate, whose count is
  // then added once)
  const uint64_t span = g_hi - g_lo, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ((span + stride - 1) / stride) * stride; t += stride) {
    const uint64_t g = g_lo + t;
    const bool live = t < span;
    const uint32_t c = live ? cid[g] : 0xFFFFFFFFu;
    bool covered = false;
    if (live && aff[c]) {
      const uint64_t first = off[c];
      for (int d = 0; d < r && !covered; d++) { if (g < first + (uint64_t)d) break; covered = hit[g - d] != 0; }
    }
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(c);
    if (__ballot(c != c0) == 0) {
      const unsigned long long m = __ballot(covered);
      if (m && (threadIdx.x & 63) == 0 && c0 != 0xFFFFFFFFu) atomicAdd(&cov[c0], (uint32_t)__popcll(m));
    } else if (covered) atomicAdd(&cov[c], 1u);
  }
}
// new decisions of the block; counts the candidates whose decision changed
__global__ void cg_decide_kernel(const unsigned long long* __restrict__ best, const uint32_t* __restrict__ cov, const uint64_t* __restrict__ off,
                                 uint32_t lo, uint32_t hi, double f, const uint8_t* __restrict__ aff, uint8_t* __restrict__ acc,
                                 uint8_t* __restrict__ chg, int32_t* __restrict__ best_count, unsigned long long* __restrict__ n_changed) {
  const uint32_t c = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= hi) return;
  if (!aff[c]) { chg[c] = 0; return; }
  const uint32_t M = (uint32_t)(best[c] >> 42);
  const bool suspect = M > 0 && (double)cov[c] > f * (double)(off[c + 1] - off[c]);
  const uint8_t a = suspect ? 0 : 1;
  best_count[c] = (int32_t)M;
  const bool changed = a != acc[c];
  chg[c] = changed ? 1 : 0;
  if (changed) { acc[c] = a; atomicAdd(n_changed, 1ULL); }
}

// ---- rounds that re-evaluate FEW candidates (round 6).  The passes above stream every accepted window of the block's index -- 1.3 +
// 0.8 + 0.4 + 0.3 ms a round at BASELINE configs[2] -- to find the windows of the handful of candidates whose view changed.  These
// forms go the other way: a workgroup per listed candidate walks the candidate's own windows (base order) and finds each window's
// place in the index through pos_of (base of the window -> its entry among the sorted shared windows) and the block's apos.
// The work a window does is the work its index entry did above, so the decisions are the same.
#define CG_NONE 0xFFFFFFFFu
__global__ void cg_pos_of_kernel(const uint32_t* __restrict__ vals, uint64_t nv, uint32_t* __restrict__ pos_of) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (uint64_t)gridDim.x * blockDim.x) pos_of[vals[i]] = (uint32_t)i;
}
// the candidates of [lo, hi) whose flag is set, in any order
__global__ void cg_list_kernel(const uint8_t* __restrict__ flag, uint32_t lo, uint32_t hi, uint32_t* __restrict__ list, uint32_t* __restrict__ n_list) {
  const uint32_t c = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (c < hi && flag[c]) list[atomicAdd(n_list, 1u)] = c;
}
// entry of window g in the block's index, or -1
__device__ __forceinline__ int64_t cg_entry(const uint32_t* __restrict__ pos_of, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ apos, uint64_t g) {
  const uint32_t i = pos_of[g];
  if (i == CG_NONE || !flag[i]) return -1;
  return (int64_t)apos[i];
}
// cg_affected_kernel for a list of changed candidates
__global__ void cg_affected_list_kernel(const uint32_t* __restrict__ clist, const uint32_t* __restrict__ n_c, const uint64_t* __restrict__ off, int r,
                                        const uint32_t* __restrict__ pos_of, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ apos,
                                        const uint64_t* __restrict__ akey, const uint32_t* __restrict__ acand, uint64_t na, uint8_t* __restrict__ aff) {
  for (uint32_t q = blockIdx.x; q < *n_c; q += gridDim.x) {
    const uint32_t d = clist[q];
    const uint64_t g0 = off[d], g1 = off[d + 1];
    if (g1 - g0 < (uint64_t)r) continue;
    for (uint64_t g = g0 + threadIdx.x; g + r <= g1; g += blockDim.x) {
      const int64_t a = cg_entry(pos_of, flag, apos, g);
      if (a < 0) continue;
      const uint64_t key = akey[a];
      for (uint64_t j = (uint64_t)a + 1; j < na && akey[j] == key; j++) { const uint32_t c = acand[j]; if (c != d) aff[c] = 1; }
    }
  }
}
// cg_clear_kernel for a list of affected candidates
__global__ void cg_clear_list_kernel(const uint32_t* __restrict__ alist, const uint32_t* __restrict__ n_a, const uint64_t* __restrict__ off,
                                     uint8_t* __restrict__ hit, unsigned long long* __restrict__ best, uint32_t* __restrict__ cov) {
  for (uint32_t q = blockIdx.x; q < *n_a; q += gridDim.x) {
    const uint32_t c = alist[q];
    for (uint64_t g = off[c] + threadIdx.x; g < off[c + 1]; g += blockDim.x) hit[g] = 0;
    if (threadIdx.x == 0) { best[c] = 0; cov[c] = 0; }
  }
}
// cg_hits_kernel for a list of affected candidates
__global__ void cg_hits_list_kernel(const uint32_t* __restrict__ alist, const uint32_t* __restrict__ n_a, const uint64_t* __restrict__ off, int r,
                                    const uint32_t* __restrict__ pos_of, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ apos,
                                    const uint64_t* __restrict__ akey, const uint32_t* __restrict__ acand, uint32_t lo, const uint8_t* __restrict__ acc,
                                    PairSlot* __restrict__ tab, uint64_t mask, uint32_t* __restrict__ overflow) {
  for (uint32_t q = blockIdx.x; q < *n_a; q += gridDim.x) {
    const uint32_t c = alist[q];
    const uint64_t g0 = off[c], g1 = off[c + 1];
    if (g1 - g0 < (uint64_t)r) continue;
    for (uint64_t g = g0 + threadIdx.x; g + r <= g1; g += blockDim.x) {
      const int64_t a = cg_entry(pos_of, flag, apos, g);
      if (a < 0) continue;
      const uint64_t key = akey[a];
      const uint32_t pos = (uint32_t)(g - g0);
      for (uint64_t j = (uint64_t)a; j-- > 0 && akey[j] == key;) {
        const uint32_t p = acand[j];
        if (p == c) continue;
        if (p >= lo && !acc[p]) continue;
        if (!pair_add(tab, mask, c, p, pos)) atomicOr(overflow, 1u);
      }
    }
  }
}
// cg_cover_kernel + cg_covsum_kernel for a list of affected candidates: the windows whose r-mer the best parent holds, then the bases
// under at least one of them
__global__ void cg_cover_list_kernel(const uint32_t* __restrict__ alist, const uint32_t* __restrict__ n_a, const uint64_t* __restrict__ off, int r,
                                     const uint32_t* __restrict__ pos_of, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ apos,
                                     const uint64_t* __restrict__ akey, const uint32_t* __restrict__ acand, const unsigned long long* __restrict__ best,
                                     uint8_t* __restrict__ hit, uint32_t* __restrict__ cov) {
  __shared__ uint32_t blk_cov;
  for (uint32_t q = blockIdx.x; q < *n_a; q += gridDim.x) {
    const uint32_t c = alist[q];
    const unsigned long long b = best[c];
    if (!b) continue;                                            // (uniform over the block: no hits, nothing covered; cov[c] was cleared)
    const uint32_t bp = (uint32_t)(b & 0x1FFFFFu);
    const uint64_t g0 = off[c], g1 = off[c + 1];
    if (threadIdx.x == 0) blk_cov = 0;
    __syncthreads();
    if (g1 - g0 >= (uint64_t)r)
      for (uint64_t g = g0 + threadIdx.x; g + r <= g1; g += blockDim.x) {
        const int64_t a = cg_entry(pos_of, flag, apos, g);
        if (a < 0) continue;
        const uint64_t key = akey[a];
        for (uint64_t j = (uint64_t)a; j-- > 0 && akey[j] == key;)
          if (acand[j] == bp) { hit[g] = 1; break; }
      }
    __syncthreads();
    uint32_t mine = 0;
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
      bool covered = false;
      for (int d = 0; d < r && !covered; d++) { if (g < g0 + (uint64_t)d) break; covered = hit[g - d] != 0; }
      mine += covered ? 1u : 0u;
    }
    if (mine) atomicAdd(&blk_cov, mine);
    __syncthreads();
    if (threadIdx.x == 0 && blk_cov) cov[c] = blk_cov;
    __syncthreads();
  }
}

// runs of equal keys that span two different contigs: flag every entry of such a run (the sort is stable and the
// windows were generated contig by contig, so a run's first and last entries differ in contig iff the run does)
__global__ void cg_shared_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n,
                                 const uint32_t* __restrict__ cid, uint32_t* __restrict__ flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i && keys[i] == keys[i - 1]) continue;                 // not a run head
    uint64_t e = i + 1;
    while (e < n && keys[e] == keys[i]) e++;
    const uint32_t f = (e - i > 1 && cid[vals[i]] != cid[vals[e - 1]]) ? 1u : 0u;
    for (uint64_t j = i; j < e; j++) flag[j] = f;
  }
}
// the same two on packed words (key << 32 | base index)
__global__ void cg_shared_packed_kernel(const uint64_t* __restrict__ words, uint64_t n, const uint32_t* __restrict__ cid, uint32_t* __restrict__ flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = words[i], key = w >> 32;
    if (i && (words[i - 1] >> 32) == key) continue;            // not a run head
    uint64_t e = i + 1, last = w;
    while (e < n) { const uint64_t x = words[e]; if ((x >> 32) != key) break; last = x; e++; }
    const uint32_t f = (e - i > 1 && cid[(uint32_t)w] != cid[(uint32_t)last]) ? 1u : 0u;
    for (uint64_t j = i; j < e; j++) flag[j] = f;
  }
}
__global__ void cg_compact_packed_kernel(const uint64_t* __restrict__ words, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ pos, uint64_t n,
                                         uint64_t* __restrict__ okeys, uint32_t* __restrict__ ovals) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (flag[i]) { const uint64_t w = words[i]; okeys[pos[i]] = w >> 32; ovals[pos[i]] = (uint32_t)w; }
}
__global__ void cg_compact_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ flag,
                                  const uint64_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ okeys, uint32_t* __restrict__ ovals) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (flag[i]) { okeys[pos[i]] = keys[i]; ovals[pos[i]] = vals[i]; }
}

static inline uint32_t grid_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cdiv(n, CG_BLK), 1), 1u << 20); }

}  // namespace

void shn_contig_ids(hipStream_t s, const uint64_t* d_off, uint64_t n_contigs, uint32_t* d_cid) {
  if (n_contigs) hipLaunchKernelGGL(cg_cid_kernel, dim3((uint32_t)cdiv(n_contigs * 64, CG_BLK)), dim3(CG_BLK), 0, s, d_off, n_contigs, d_cid);
}

// sorted (key, base index) pairs of all k-windows of the selected contigs; device arrays in *keys / *vals (owned by bufs)
int shn_sorted_windows(shn_ctx* ctx, ShnDevBufs& bufs, const uint8_t* d_bases, const uint64_t* d_off, const uint32_t* d_cid,
                          const int32_t* d_use, uint64_t total, int k, uint64_t** keys, uint32_t** vals, uint64_t* n_out) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint32_t* d_flag; uint64_t* d_pos; unsigned long long* d_bad;
  HIP_TRY(bufs.get(&d_flag, (total + 1) * 4));
  HIP_TRY(bufs.get(&d_pos, (total + 2) * 8));
  HIP_TRY(bufs.get(&d_bad, 8));
  HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
  hipLaunchKernelGGL(cg_flag_kernel, dim3(grid_for(total)), dim3(CG_BLK), 0, s, d_bases, d_off, d_cid, d_use, total, k, d_flag);
  uint64_t nv = 0;
  int rc = shn_device_scan_u32(ctx, d_flag, total, d_pos, &nv);
  if (rc) return rc;
  *n_out = nv;
  *keys = nullptr; *vals = nullptr;
  if (!nv) return SHN_OK;
  uint64_t *k1, *k2; uint32_t *v1, *v2;
  HIP_TRY(bufs.get(&k1, nv * 8)); HIP_TRY(bufs.get(&k2, nv * 8)); HIP_TRY(bufs.get(&v1, nv * 4)); HIP_TRY(bufs.get(&v2, nv * 4));
  hipLaunchKernelGGL(cg_keys_kernel, dim3(grid_for(total)), dim3(CG_BLK), 0, s, d_bases, d_flag, d_pos, total, k, k1, v1, d_bad);
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (bad) return shn_fail(SHN_ERR_ARG, "shn_contig_stage: contig with a base outside ACGT");
  if ((rc = shn_sort_pairs(ctx, k1, v1, k2, v2, nv, 0, 2 * k))) return rc;
  *keys = k1; *vals = v1;
  return SHN_OK;
}

// k <= 16: the same windows as words (key << 32 | base index), sorted by key, stable
static int sorted_windows_packed(shn_ctx* ctx, ShnDevBufs& bufs, const uint8_t* d_bases, const uint64_t* d_off, const uint32_t* d_cid, uint64_t total, int k,
                                 uint64_t** words, uint64_t* n_out) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint32_t* d_flag; uint64_t* d_pos; unsigned long long* d_bad;
  HIP_TRY(bufs.get(&d_flag, (total + 1) * 4));
  HIP_TRY(bufs.get(&d_pos, (total + 2) * 8));
  HIP_TRY(bufs.get(&d_bad, 8));
  HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
  hipLaunchKernelGGL(cg_flag_kernel, dim3(grid_for(total)), dim3(CG_BLK), 0, s, d_bases, d_off, d_cid, (const int32_t*)nullptr, total, k, d_flag);
  uint64_t nv = 0;
  int rc = shn_device_scan_u32(ctx, d_flag, total, d_pos, &nv);
  if (rc) return rc;
  *n_out = nv;
  *words = nullptr;
  if (!nv) return SHN_OK;
  uint64_t *w1, *w2;
  HIP_TRY(bufs.get(&w1, nv * 8)); HIP_TRY(bufs.get(&w2, nv * 8));
  hipLaunchKernelGGL(cg_keys_packed_kernel, dim3(grid_for(total)), dim3(CG_BLK), 0, s, d_bases, d_flag, d_pos, total, k, w1, d_bad);
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (bad) return shn_fail(SHN_ERR_ARG, "shn_contig_stage: contig with a base outside ACGT");
  return shn_sort_keys(ctx, w1, w2, nv, 32, 32 + 2 * k, words);
}

// Same contract as shn_cgraph_add on a fresh graph: candidates in seed order in one call; accepted_out[i] = 1-based accepted
// index or 0, best_counts_out[i] (may be NULL) = hit count of the candidate's `best` contig; *out receives the contig graph
// (shn_cgraph_sizes / shn_cgraph_export / shn_cgraph_destroy).
// bases: the candidates' text on the host (uploaded), or dev_text: the same text already on the device (>= total + 64 bytes)
static int contig_stage_impl(shn_ctx* ctx, const uint8_t* bases, const uint8_t* dev_text, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                             int32_t* accepted_out, int32_t* best_counts_out, shn_cgraph** out) {
  if (!ctx || !out || k1 < 2 || k1 > 33 || r < 1 || r > 32 || (n_cand && ((!bases && !dev_text) || !off || !accepted_out)))
    return shn_fail(SHN_ERR_ARG, "shn_contig_stage: bad argument");
  *out = nullptr;
  const bool dbg = getenv("SHN_DEBUG") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char* what) { if (dbg) { double t = now(); fprintf(stderr, "[contig_stage] %-34s %8.3f s\n", what, t - t0); t0 = t; } };
  shn_cgraph* G = new shn_cgraph(k1, r, f);
  struct Guard { shn_cgraph* g; ~Guard() { delete g; } } guard{G};
  std::vector<int32_t> best_tmp;
  if (!best_counts_out) { best_tmp.assign(n_cand + 1, 0); best_counts_out = best_tmp.data(); }
  for (uint64_t i = 0; i < n_cand; i++) { accepted_out[i] = 0; best_counts_out[i] = 0; }
  const uint64_t total = n_cand ? off[n_cand] - off[0] : 0;
  if (!n_cand || !total) { *out = G; guard.g = nullptr; return SHN_OK; }
  if (off[0] != 0) return shn_fail(SHN_ERR_ARG, "shn_contig_stage: offsets must start at 0");
  if (total >= 0xFFFFFFF0ULL || n_cand >= 0x7FFFFFF0ULL) return shn_fail(SHN_ERR_OVERFLOW, "shn_contig_stage: more than 2^32 contig bases");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  shn_stage_begin(ctx);
  TimerRegion treg(ctx, T_CONTIG);
  ShnDevBufs bufs(s);
  uint8_t* d_bases; uint64_t* d_off; uint32_t* d_cid;
  if (dev_text) d_bases = const_cast<uint8_t*>(dev_text);            // (read only from here on)
  else HIP_TRY(bufs.get(&d_bases, total + 64));
  HIP_TRY(bufs.get(&d_off, (n_cand + 1) * 8));
  HIP_TRY(bufs.get(&d_cid, (total + 1) * 4));
  if (!dev_text) HIP_TRY(hipMemcpyAsync(d_bases, bases, total, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_off, off, (n_cand + 1) * 8, hipMemcpyHostToDevice, s));
  shn_contig_ids(s, d_off, n_cand, d_cid);

  // ---- duplicate_check: seed-order blocks, each iterated to its fixpoint on top of the frozen earlier blocks
  std::vector<uint8_t> acc(n_cand, 0);
  uint32_t max_len = 0;
  for (uint64_t c = 0; c < n_cand; c++) max_len = std::max<uint32_t>(max_len, (uint32_t)(off[c + 1] - off[c]));
  if (n_cand >= (1u << 21) || max_len >= (1u << 21)) {
    // beyond the 21-bit fields of the packed "best parent" word: the sequential stage
    std::vector<uint8_t> fetched;
    if (!bases) {
      fetched.resize(total);
      HIP_TRY(hipMemcpyAsync(fetched.data(), d_bases, total, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      bases = fetched.data();
    }
    G->add(bases, off, n_cand, accepted_out, best_counts_out);
    *out = G; guard.g = nullptr;
    return SHN_OK;
  }
  uint64_t n_rounds = 0, n_blocks = 0, n_list_rounds = 0;
  {
    ShnDevBufs tmp(s);
    uint64_t* keys = nullptr; uint32_t* vals = nullptr; uint64_t* words = nullptr; uint64_t nv = 0;
    int rc;
    const bool shared_only = !(getenv("SHN_CONTIG_SHARED") && getenv("SHN_CONTIG_SHARED")[0] == '0');
    // (r <= 16: the r-mer and the 32-bit base index of its window share one word through the sort; SHN_CONTIG_PACKED=0: pairs)
    const bool packed = r <= 16 && shared_only && !(getenv("SHN_CONTIG_PACKED") && getenv("SHN_CONTIG_PACKED")[0] == '0');
    { TimerRegion ts(ctx, T_CG_SORT);
      rc = packed ? sorted_windows_packed(ctx, tmp, d_bases, d_off, d_cid, total, r, &words, &nv)
                  : shn_sorted_windows(ctx, tmp, d_bases, d_off, d_cid, nullptr, total, r, &keys, &vals, &nv);
      // per r-mer window of the candidates: its base read, (key 8 + value 4, or one word of 8) written, then an LSD pass per 8 key bits
      // that reads and writes the entry
      ts.bytes(packed ? nv * (1 + 8 + (uint64_t)((2 * r + 7) / 8) * 16) : nv * (1 + 12 + (uint64_t)((2 * r + 7) / 8) * 24)); }
    if (rc) return rc;
    lap("r-mer sort (GPU)");
    // Only the runs that span two different candidates matter from here on: a window whose r-mer no OTHER candidate holds has no hit
    // (:250-259 counts the earlier candidates sharing it), covers nothing, affects nobody -- and three windows out of four are such
    // (4^15 possible 15-mers against 3 x 10^8 windows at BASELINE configs[2]: a quarter of the windows collide by chance alone).  The
    // per-block index and every round's passes (hits, cover, affected) then stream a quarter of the entries.  The same flag and
    // compaction as for the K-mer join below (stable sort + windows made candidate by candidate: a run spans two candidates iff
    // its first and last entries differ in candidate).  SHN_CONTIG_SHARED=0: all windows, as until round 6.
    if (nv > 1 && shared_only) {
      uint32_t* d_f0; uint64_t* d_p0; uint64_t ns0 = 0;
      ShnDevBufs scratch(s);
      HIP_TRY(scratch.get(&d_f0, (nv + 1) * 4));
      HIP_TRY(scratch.get(&d_p0, (nv + 2) * 8));
      if (packed) hipLaunchKernelGGL(cg_shared_packed_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, words, nv, d_cid, d_f0);
      else hipLaunchKernelGGL(cg_shared_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, nv, d_cid, d_f0);
      if ((rc = shn_device_scan_u32(ctx, d_f0, nv, d_p0, &ns0))) return rc;
      uint64_t* ok = nullptr; uint32_t* ov = nullptr;
      HIP_TRY(tmp.get(&ok, (ns0 + 1) * 8)); HIP_TRY(tmp.get(&ov, (ns0 + 1) * 4));
      if (ns0 && packed) hipLaunchKernelGGL(cg_compact_packed_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, words, d_f0, d_p0, nv, ok, ov);
      else if (ns0) hipLaunchKernelGGL(cg_compact_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, d_f0, d_p0, nv, ok, ov);
      if (dbg) fprintf(stderr, "[contig_stage]   %llu of %llu r-mer windows lie in runs that span two candidates\n", (unsigned long long)ns0, (unsigned long long)nv);
      keys = ok; vals = ov; nv = ns0;
      lap("shared r-mer runs (GPU)");
    } else if (packed && nv == 1) {
      // (one window: no run to share -- the rounds below see an empty index)
      nv = 0;
    }
    uint8_t *d_acc, *d_hit; uint32_t *d_scid, *d_flag, *d_acand, *d_cov, *d_ovf; uint64_t *d_apos, *d_akey; int32_t* d_bestc;
    unsigned long long *d_best, *d_chg;
    HIP_TRY(tmp.get(&d_acc, n_cand + 1)); HIP_TRY(tmp.get(&d_hit, total + 64)); HIP_TRY(tmp.get(&d_scid, (nv + 1) * 4));
    HIP_TRY(tmp.get(&d_flag, (nv + 1) * 4)); HIP_TRY(tmp.get(&d_acand, (nv + 1) * 4)); HIP_TRY(tmp.get(&d_cov, (n_cand + 1) * 4));
    HIP_TRY(tmp.get(&d_ovf, 64)); HIP_TRY(tmp.get(&d_apos, (nv + 2) * 8)); HIP_TRY(tmp.get(&d_akey, (nv + 1) * 8));
    HIP_TRY(tmp.get(&d_bestc, (n_cand + 1) * 4)); HIP_TRY(tmp.get(&d_best, (n_cand + 1) * 8)); HIP_TRY(tmp.get(&d_chg, 64));
    uint8_t *d_aff, *d_chgf; uint32_t* d_aval;
    HIP_TRY(tmp.get(&d_aval, (nv + 1) * 4));
    HIP_TRY(tmp.get(&d_aff, n_cand + 1)); HIP_TRY(tmp.get(&d_chgf, n_cand + 1));
    HIP_TRY(hipMemsetAsync(d_chgf, 0, n_cand + 1, s));
    const bool incremental = !getenv("SHN_CONTIG_INCREMENTAL") || atoi(getenv("SHN_CONTIG_INCREMENTAL")) != 0;
    HIP_TRY(hipMemsetAsync(d_acc, 0, n_cand + 1, s));
    HIP_TRY(hipMemsetAsync(d_bestc, 0, (n_cand + 1) * 4, s));
    if (nv) hipLaunchKernelGGL(cg_scid_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, vals, d_cid, nv, d_scid);
    // rounds that re-evaluate few candidates go candidate by candidate (cg_*_list_kernel): base of a window -> its sorted entry
    const uint64_t list_max = getenv("SHN_CONTIG_LIST_MAX") ? strtoull(getenv("SHN_CONTIG_LIST_MAX"), nullptr, 10) : 20000;      // decisions changed in the round before (0: never)
    uint32_t *d_pos_of = nullptr, *d_clist = nullptr, *d_alist = nullptr, *d_nlist = nullptr;
    if (list_max && nv && nv < 0xFFFFFFF0ULL) {
      HIP_TRY(tmp.get(&d_pos_of, (total + 1) * 4)); HIP_TRY(tmp.get(&d_clist, (n_cand + 1) * 4)); HIP_TRY(tmp.get(&d_alist, (n_cand + 1) * 4)); HIP_TRY(tmp.get(&d_nlist, 64));
      HIP_TRY(hipMemsetAsync(d_pos_of, 0xFF, (total + 1) * 4, s));
      hipLaunchKernelGGL(cg_pos_of_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, vals, nv, d_pos_of);
    }
    int lg_slots = 22;
    while (lg_slots < 28 && (1ULL << lg_slots) < nv / 4) lg_slots++;
    if (getenv("SHN_CONTIG_PAIR_LOG2")) lg_slots = atoi(getenv("SHN_CONTIG_PAIR_LOG2"));     // (tests: start too small, grow)
    PairSlot* d_tab = nullptr;
    HIP_TRY(tmp.get(&d_tab, sizeof(PairSlot) << lg_slots));
    const int lg_small = getenv("SHN_CONTIG_PAIR_SMALL_LOG2") ? atoi(getenv("SHN_CONTIG_PAIR_SMALL_LOG2")) : 20;   // pair table of the rounds after a block's first
    uint64_t blk0 = std::max<uint64_t>(1024, n_cand / 16);          // (n / 256: 23 rounds, 0.75 s at 731 k candidates; n / 16: 0.66 s; one block: 0.68 s)
    if (getenv("SHN_CONTIG_BLOCK0")) blk0 = std::max<uint64_t>(1, strtoull(getenv("SHN_CONTIG_BLOCK0"), nullptr, 10));
    const int max_rounds = getenv("SHN_CONTIG_MAX_ROUNDS") ? std::max(1, atoi(getenv("SHN_CONTIG_MAX_ROUNDS"))) : 64;   // (tests: halve blocks early)
    uint64_t lo = 0, bsize = blk0;
    while (lo < n_cand) {
      uint64_t hi = std::min<uint64_t>(n_cand, lo + bsize);
      // the candidates of the block start as "not accepted": the first round meets the frozen earlier blocks only
      int round = 0;
      uint64_t na = 0;
      unsigned long long last_changed = 0;
      if (nv) {
        hipLaunchKernelGGL(cg_accflag_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, d_scid, d_acc, nv, (uint32_t)lo, (uint32_t)hi, d_flag);
        if ((rc = shn_device_scan_u32(ctx, d_flag, nv, d_apos, &na))) return rc;
        if (na) { TimerRegion tc(ctx, T_CG_COMPACT); tc.bytes(nv * 28 + na * 16);      // every window: key 8 + candidate 4 + flag 4 + value 4 + position 8 read; the accepted ones written (16)
                  hipLaunchKernelGGL(cg_acc_compact_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, d_scid, d_flag, vals, d_apos, nv, d_akey, d_acand, d_aval); }
      }
      HIP_TRY(hipMemsetAsync(d_aff + lo, 1, hi - lo, s));
      while (true) {
        unsigned long long changed = 0;
        const bool by_list = round > 0 && incremental && d_pos_of && na && last_changed <= list_max;
        const uint32_t lgrid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, last_changed * 8), 16384);
        if (round == 0 || !incremental) {
          HIP_TRY(hipMemsetAsync(d_best + lo, 0, (hi - lo) * 8, s));
          HIP_TRY(hipMemsetAsync(d_cov + lo, 0, (hi - lo) * 4, s));
          HIP_TRY(hipMemsetAsync(d_hit + off[lo], 0, off[hi] - off[lo], s));
        } else if (by_list) {
          n_list_rounds++;
          HIP_TRY(hipMemsetAsync(d_aff + lo, 0, hi - lo, s));
          HIP_TRY(hipMemsetAsync(d_nlist, 0, 8, s));
          hipLaunchKernelGGL(cg_list_kernel, dim3((uint32_t)cdiv(hi - lo, CG_BLK)), dim3(CG_BLK), 0, s, d_chgf, (uint32_t)lo, (uint32_t)hi, d_clist, d_nlist);
          hipLaunchKernelGGL(cg_affected_list_kernel, dim3(lgrid), dim3(CG_BLK), 0, s, d_clist, d_nlist, d_off, r, d_pos_of, d_flag, d_apos, d_akey, d_acand, na, d_aff);
          hipLaunchKernelGGL(cg_list_kernel, dim3((uint32_t)cdiv(hi - lo, CG_BLK)), dim3(CG_BLK), 0, s, d_aff, (uint32_t)lo, (uint32_t)hi, d_alist, d_nlist + 1);
          hipLaunchKernelGGL(cg_clear_list_kernel, dim3(lgrid), dim3(CG_BLK), 0, s, d_alist, d_nlist + 1, d_off, d_hit, d_best, d_cov);
        } else {
          HIP_TRY(hipMemsetAsync(d_aff + lo, 0, hi - lo, s));
          hipLaunchKernelGGL(cg_affected_kernel, dim3(grid_for(na)), dim3(CG_BLK), 0, s, d_akey, d_acand, na, (uint32_t)lo, d_chgf, d_aff);
          hipLaunchKernelGGL(cg_clear_kernel, dim3(grid_for(off[hi] - off[lo])), dim3(CG_BLK), 0, s, d_aff, d_cid, off[lo], off[hi], (uint32_t)lo, (uint32_t)hi,
                             d_hit, d_best, d_cov);
        }
        HIP_TRY(hipMemsetAsync(d_chg, 0, 8, s));
        if (na) {
          // (a round after a block's first re-evaluates the few candidates whose view changed -- tens to thousands of 10^5: its pairs
          // fit a table a hundredth the size, and clearing and scanning the full 2 GB table was most of such a round)
          const bool small_round = round > 0 && incremental;
          int lg_use = lg_slots;
          if (small_round) {                         // (room for 4,096 pairs per decision that changed in the round before; at least 2^lg_small slots)
            lg_use = lg_small;
            while (lg_use < lg_slots && (1ULL << lg_use) < last_changed * 4096ULL) lg_use++;
            lg_use = std::min(lg_use, lg_slots);
          }
          while (true) {                             // (the pair table grows until the round's pairs fit)
            HIP_TRY(hipMemsetAsync(d_tab, 0, sizeof(PairSlot) << lg_use, s));
            HIP_TRY(hipMemsetAsync(d_ovf, 0, 4, s));
            if (by_list)
              hipLaunchKernelGGL(cg_hits_list_kernel, dim3(lgrid), dim3(CG_BLK), 0, s, d_alist, d_nlist + 1, d_off, r, d_pos_of, d_flag, d_apos, d_akey, d_acand,
                                 (uint32_t)lo, d_acc, d_tab, (1ULL << lg_use) - 1, d_ovf);
            else
            { TimerRegion th(ctx, T_CG_HITS); th.bytes(na * 16);                     // every accepted window: key 8 + candidate 4 + value 4 (the pair table: a few per cent of them)
              hipLaunchKernelGGL(cg_hits_kernel, dim3(grid_for(na)), dim3(CG_BLK), 0, s, d_akey, d_acand, d_aval, na, d_off, (uint32_t)lo, d_acc, d_aff,
                                 d_tab, (1ULL << lg_use) - 1, d_ovf); }
            uint32_t ovf = 0;
            HIP_TRY(hipMemcpyAsync(&ovf, d_ovf, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (!ovf) break;
            if (lg_use < lg_slots) { lg_use = lg_slots; continue; }                          // (the whole table that is there)
            if (lg_slots >= 31) return shn_fail(SHN_ERR_OVERFLOW, "shn_contig_stage: pair table beyond 2^31 slots");
            lg_slots += 2;
            lg_use = lg_slots;
            HIP_TRY(tmp.get(&d_tab, sizeof(PairSlot) << lg_slots));
          }
          hipLaunchKernelGGL(cg_best_kernel, dim3(grid_for(1ULL << lg_use)), dim3(CG_BLK), 0, s, d_tab, 1ULL << lg_use, d_best);
          if (by_list)
            hipLaunchKernelGGL(cg_cover_list_kernel, dim3(lgrid), dim3(CG_BLK), 0, s, d_alist, d_nlist + 1, d_off, r, d_pos_of, d_flag, d_apos, d_akey, d_acand, d_best,
                               d_hit, d_cov);
          else {
          { TimerRegion tv(ctx, T_CG_COVER); tv.bytes(na * 17);                       // the same 16 bytes + the hit byte of the window's base
            hipLaunchKernelGGL(cg_cover_kernel, dim3(grid_for(na)), dim3(CG_BLK), 0, s, d_akey, d_acand, d_aval, na, (uint32_t)lo, d_best, d_aff, d_hit); }
          hipLaunchKernelGGL(cg_covsum_kernel, dim3(grid_for(off[hi] - off[lo])), dim3(CG_BLK), 0, s, d_hit, d_cid, d_off, off[lo], off[hi], r, d_aff, d_cov);
          }
        }
        hipLaunchKernelGGL(cg_decide_kernel, dim3((uint32_t)cdiv(hi - lo, CG_BLK)), dim3(CG_BLK), 0, s, d_best, d_cov, d_off, (uint32_t)lo, (uint32_t)hi, f,
                           d_aff, d_acc, d_chgf, d_bestc, d_chg);
        HIP_TRY(hipMemcpyAsync(&changed, d_chg, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        n_rounds++;
        round++;
        last_changed = changed;
        if (dbg) fprintf(stderr, "[contig_stage]   block [%llu,%llu) round %d: %llu accepted entries, %llu decisions changed\n", (unsigned long long)lo,
                         (unsigned long long)hi, round, (unsigned long long)na, changed);
        if (!changed) break;
        if (round >= max_rounds && hi - lo > 1) {
          // a long dependency chain inside the block: go on with its first half (its fixpoint does not depend on the rest);
          // the candidates cut off go back to "not accepted" and come with the next block
          HIP_TRY(hipMemsetAsync(d_aff + lo + (hi - lo) / 2, 0, hi - (lo + (hi - lo) / 2), s));
          hi = lo + (hi - lo) / 2;
          HIP_TRY(hipMemsetAsync(d_acc + hi, 0, n_cand - hi, s));
          HIP_TRY(hipMemsetAsync(d_aff + lo, 1, hi - lo, s));         // (everything of the half is looked at again: the candidates cut off were in its hits)
          round = 0;
        }
      }
      n_blocks++;
      lo = hi;
      bsize = std::min<uint64_t>(bsize * 2, std::max<uint64_t>(blk0, n_cand / 4));
    }
    HIP_TRY(hipMemcpyAsync(acc.data(), d_acc, n_cand, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(best_counts_out, d_bestc, n_cand * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  std::vector<int32_t> use(n_cand);
  int32_t n_acc = 0;
  for (uint64_t c = 0; c < n_cand; c++) { use[c] = acc[c] ? ++n_acc : 0; accepted_out[c] = use[c]; }
  if (dbg) fprintf(stderr, "[contig_stage] %llu candidates (%llu bases): %llu blocks, %llu rounds (%llu of them candidate by candidate), accepted %d\n",
                   (unsigned long long)n_cand, (unsigned long long)total, (unsigned long long)n_blocks, (unsigned long long)n_rounds, (unsigned long long)n_list_rounds, n_acc);
  lap("duplicate_check rounds (GPU)");

  // ---- contig_connections: K-mers occurring in two different accepted contigs
  G->idx = n_acc;
  G->conns.assign((size_t)n_acc + 1, Conn());
  G->n_cand_total = n_cand;
  const int C = k1 - 1;
  if (n_acc > 1) {
    int32_t* d_use;
    HIP_TRY(bufs.get(&d_use, n_cand * 4));
    HIP_TRY(hipMemcpyAsync(d_use, use.data(), n_cand * 4, hipMemcpyHostToDevice, s));
    ShnDevBufs tmp(s);
    uint64_t* keys; uint32_t* vals; uint64_t nv = 0;
    int rc = shn_sorted_windows(ctx, tmp, d_bases, d_off, d_cid, d_use, total, C, &keys, &vals, &nv);
    if (rc) return rc;
    uint64_t ns = 0;
    std::vector<uint64_t> hk; std::vector<uint32_t> hg;
    if (nv > 1) {
      uint32_t* d_flag; uint64_t* d_pos;
      HIP_TRY(tmp.get(&d_flag, (nv + 1) * 4));
      HIP_TRY(tmp.get(&d_pos, (nv + 2) * 8));
      hipLaunchKernelGGL(cg_shared_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, nv, d_cid, d_flag);
      if ((rc = shn_device_scan_u32(ctx, d_flag, nv, d_pos, &ns))) return rc;
      if (ns) {
        uint64_t* ok; uint32_t* ov;
        HIP_TRY(tmp.get(&ok, ns * 8)); HIP_TRY(tmp.get(&ov, ns * 4));
        hipLaunchKernelGGL(cg_compact_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, d_flag, d_pos, nv, ok, ov);
        hk.resize(ns); hg.resize(ns);
        HIP_TRY(hipMemcpyAsync(hk.data(), ok, ns * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(hg.data(), ov, ns * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
      }
    }
    if (dbg) fprintf(stderr, "[contig_stage]   %llu of %llu K-mer windows of the accepted contigs lie in runs that span two of them\n", (unsigned long long)ns, (unsigned long long)nv);
    lap("K-mer sort + shared runs (GPU)");
    // host replay of the reference's loop over the shared runs only.  Entry e: K-mer run `run[e]`, accepted contig `ea[e]`,
    // position `ep[e]`; inside a run the entries are in (contig, position) order = the order of cmer_to_contig's list.
    if (ns) {
      std::vector<uint32_t> run(ns), ea(ns), ep(ns), run_start;
      for (uint64_t e = 0; e < ns; e++) {
        if (e == 0 || hk[e] != hk[e - 1]) run_start.push_back((uint32_t)e);
        run[e] = (uint32_t)run_start.size() - 1;
      }
      run_start.push_back((uint32_t)ns);
      // the shared windows of every accepted contig in position order = the entries by their base g in the candidates' text
      // (accepted contigs are numbered in candidate order): three counting passes over 11 bits of g, then one walk along the
      // offsets names every entry's contig and position (a comparison sort of the entries and a binary search per entry were two
      // thirds of this replay)
      std::vector<uint32_t> order(ns), order2(ns);
      {
        for (uint64_t e = 0; e < ns; e++) order[e] = (uint32_t)e;
        std::vector<uint32_t> cnt(2049);
        for (int pass = 0; pass < 3; pass++) {
          const int sh = 11 * pass;
          if (sh && (total >> sh) == 0) break;
          std::fill(cnt.begin(), cnt.end(), 0u);
          for (uint64_t e = 0; e < ns; e++) cnt[((hg[e] >> sh) & 2047) + 1]++;
          for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
          for (uint64_t i = 0; i < ns; i++) { const uint32_t e = order[i]; order2[cnt[(hg[e] >> sh) & 2047]++] = e; }
          order.swap(order2);
        }
        uint64_t c = 0;
        for (uint64_t i = 0; i < ns; i++) {
          const uint32_t e = order[i];
          const uint64_t g = hg[e];
          while (off[c + 1] <= g) c++;
          ea[e] = (uint32_t)use[c];
          ep[e] = (uint32_t)(g - off[c]);
        }
      }
      std::vector<int32_t> connw((size_t)n_acc + 1, 0), newnb;
      size_t q = 0;
      while (q < ns) {
        const uint32_t a = ea[order[q]];
        newnb.clear();
        for (; q < ns && ea[order[q]] == a; q++) {
          const uint32_t rn = run[order[q]];
          for (uint32_t e = run_start[rn]; e < run_start[rn + 1]; e++) {
            const int32_t c2 = (int32_t)ea[e];
            if (c2 >= (int32_t)a) break;                 // later contigs are not in the index yet; own occurrences are skipped
            if (connw[c2]++ == 0) newnb.push_back(c2);
            Conn& b = G->conns[c2];
            if (!b.nb.empty() && b.nb.back() == (int32_t)a) b.w.back()++;
            else { b.nb.push_back((int32_t)a); b.w.push_back(1); }
          }
        }
        // contig `a` was accepted before any later contig touched its list: its own neighbours come first
        Conn& A = G->conns[a];
        std::vector<int32_t> nb(newnb), w(newnb.size());
        for (size_t j = 0; j < newnb.size(); j++) { w[j] = connw[newnb[j]]; connw[newnb[j]] = 0; }
        nb.insert(nb.end(), A.nb.begin(), A.nb.end());
        w.insert(w.end(), A.w.begin(), A.w.end());
        A.nb.swap(nb); A.w.swap(w);
      }
    }
    lap("contig_connections replay (host)");
  }
  HIP_TRY(hipGetLastError());
  *out = G;
  guard.g = nullptr;
  return SHN_OK;
}

extern "C" int shn_contig_stage(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                                int32_t* accepted_out, int32_t* best_counts_out, shn_cgraph** out) {
  if (n_cand && !bases) return shn_fail(SHN_ERR_ARG, "shn_contig_stage: bad argument");
  return contig_stage_impl(ctx, bases, nullptr, off, n_cand, k1, r, f, accepted_out, best_counts_out, out);
}
// The candidates' text where shn_ext_emit_device left it: no copy of the 0.3 GB of candidate contigs to the host and back (only the
// accepted tenth is ever needed there: shn_devtext_segments).
extern "C" int shn_contig_stage_device(shn_ctx* ctx, const shn_devtext* text, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                                       int32_t* accepted_out, int32_t* best_counts_out, shn_cgraph** out) {
  if (n_cand && (!text || !text->d || !off || text->n < off[n_cand])) return shn_fail(SHN_ERR_ARG, "shn_contig_stage_device: the text does not cover the offsets");
  return contig_stage_impl(ctx, nullptr, text ? text->d : nullptr, off, n_cand, k1, r, f, accepted_out, best_counts_out, out);
}
extern "C" void shn_devtext_destroy(shn_devtext* t) {
  if (!t) return;
  if (t->ctx) hipSetDevice(t->ctx->device);
  shn_dev_free(t->d);
  delete t;
}
__global__ void devtext_segments_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ seg /* src begin, dst begin per segment; [n] = total */,
                                        uint64_t n_seg, uint64_t total, uint8_t* __restrict__ dst) {
  // one thread per output byte: its segment by bisection over the destination offsets
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t lo = 0, hi = n_seg;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (seg[2 * mid + 1] <= g) lo = mid; else hi = mid; }
    dst[g] = src[seg[2 * lo] + (g - seg[2 * lo + 1])];
  }
}
// segments idx[0..n_idx) (of the n_off - 1 pieces the offsets `off` cut the text into) one after the other into out (host)
extern "C" int shn_devtext_segments(shn_ctx* ctx, const shn_devtext* text, const uint64_t* off, uint64_t n_off, const int64_t* idx, uint64_t n_idx, uint8_t* out) {
  if (!ctx || !text || (n_idx && (!off || !idx || !out))) return shn_fail(SHN_ERR_ARG, "shn_devtext_segments: NULL argument");
  if (!n_idx) return SHN_OK;
  std::vector<uint64_t> seg(2 * n_idx + 2);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n_idx; i++) {
    if (idx[i] < 0 || (uint64_t)idx[i] + 1 >= n_off || off[idx[i] + 1] < off[idx[i]] || off[idx[i] + 1] > text->n) return shn_fail(SHN_ERR_ARG, "shn_devtext_segments: segment outside the text");
    seg[2 * i] = off[idx[i]]; seg[2 * i + 1] = total;
    total += off[idx[i] + 1] - off[idx[i]];
  }
  seg[2 * n_idx] = 0; seg[2 * n_idx + 1] = total;
  if (!total) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  ShnDevBufs bufs(s);
  uint64_t* d_seg; uint8_t* d_dst;
  HIP_TRY(bufs.get(&d_seg, seg.size() * 8)); HIP_TRY(bufs.get(&d_dst, total));
  HIP_TRY(hipMemcpyAsync(d_seg, seg.data(), seg.size() * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(devtext_segments_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(total, 256), 1u << 16)), dim3(256), 0, s, (const uint8_t*)text->d, (const uint64_t*)d_seg,
                     n_idx, total, d_dst);
  HIP_TRY(hipMemcpyAsync(out, d_dst, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}
