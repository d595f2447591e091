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
  uint32_t* d_claim;     // [2n] converged claims (rank of the walk owning each oriented k1-mer)
  uint32_t* d_claim2;    // [2n] scratch
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

// One thread per walk.  EMIT: write the contig bases (ASCII) of the selected walks.
// Per step ONE round of independent loads: for each of the 4 candidates (ids already in registers)
// its claim (previous iteration + final), its claim in this iteration, its weight and its own
// adjacency row (prefetched, so the next step needs no dependent load); the claim is published
// with a fire-and-forget atomicMin.  The step latency is one L2/HBM round trip, not seven.
struct Adj4 { int32_t v[4]; };

template <bool EMIT>
__global__ __launch_bounds__(EBLK) void ext_walk_kernel(const uint32_t* __restrict__ order, uint64_t n_walks,
                                                        const uint32_t* __restrict__ sel,   // EMIT: ranks to emit
                                                        const int32_t* __restrict__ adjR, const int32_t* __restrict__ adjL,
                                                        const uint32_t* __restrict__ weight, const uint32_t* __restrict__ claim_prev,
                                                        uint32_t first, uint32_t* __restrict__ claim_cur,
                                                        uint32_t* __restrict__ nr_out, uint32_t* __restrict__ nl_out,
                                                        uint64_t* __restrict__ totw_out, uint64_t* __restrict__ hash_io,
                                                        uint32_t* __restrict__ changed, const uint64_t* __restrict__ tkeys, int k,
                                                        const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out_bases) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_walks) return;
  const uint32_t r = EMIT ? sel[t] : (uint32_t)t + first;
  const uint32_t o = order[r];
  uint32_t nr = 0, nl = 0;
  uint64_t tot = 0, h = 0x9E3779B97F4A7C15ULL;
  // claim_prev holds the claims of the previous iteration merged with all final claims
  bool isvoid = claim_prev[o] < r;
  uint8_t* dst = nullptr;
  uint32_t nl_known = 0;
  if (EMIT) {
    if (isvoid) return;
    dst = out_bases + out_off[t];
    nl_known = nl_out[r];
    uint64_t s = oriented_string(tkeys, o, k);
    for (int j = 0; j < k; j++) dst[nl_known + j] = "ACGT"[(s >> (2 * (k - 1 - j))) & 3];
  }
  if (!isvoid) {
    atomicMin(&claim_cur[o], r);
    tot = weight[o >> 1];
    for (int dir = 0; dir < 2; dir++) {
      const Adj4* adj = (const Adj4*)(dir == 0 ? adjR : adjL);
      uint32_t steps = 0;
      Adj4 cand = adj[o];
      while (true) {
        uint32_t cp[4], cc[4], w[4];
        Adj4 nxt[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          int32_t nb = cand.v[b];
          uint32_t idx = nb < 0 ? o : (uint32_t)nb;        // harmless address for absent candidates
          cp[b] = claim_prev[idx];
          cc[b] = __hip_atomic_load(&claim_cur[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w[b] = weight[idx >> 1];
          nxt[b] = adj[idx];
        }
        int best = -1;
        uint32_t bw = 0;
        // BASES = ['A','G','C','T'] (extension_correction.py:10): codes 0,2,1,3; strict > keeps the first
#define CONSIDER(b) if (cand.v[b] >= 0 && cp[b] >= r && cc[b] > r && (best < 0 || w[b] > bw)) { best = b; bw = w[b]; }
        CONSIDER(0) CONSIDER(2) CONSIDER(1) CONSIDER(3)
#undef CONSIDER
        if (best < 0) break;
        uint32_t nbest = (uint32_t)cand.v[best];
        __hip_atomic_fetch_min(&claim_cur[nbest], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (EMIT) {
          if (dir == 0) dst[nl_known + k + steps] = "ACGT"[best];
          else dst[nl_known - 1 - steps] = "ACGT"[best];
        }
        steps++;
        tot += bw;
        h = shn_mix64(h ^ (uint64_t)nbest);
        cand = best == 0 ? nxt[0] : best == 1 ? nxt[1] : best == 2 ? nxt[2] : nxt[3];
      }
      if (dir == 0) nr = steps; else nl = steps;
    }
  }
  if (!EMIT) {
    uint64_t hh = isvoid ? 0ULL : (h | 1ULL);
    if (hash_io[r] != hh) { hash_io[r] = hh; atomicMin(changed, r); }   // changed = lowest rank whose path changed
    nr_out[r] = isvoid ? UNCLAIMED : nr;
    nl_out[r] = nl;
    totw_out[r] = tot;
  }
}

// Freeze the claims of the walks that just became final into `fin`, and merge every final claim into
// `cur`, which is the next iteration's claim_prev (final claims have rank < first <= any live walk).
__global__ void ext_freeze_kernel(uint32_t* __restrict__ cur, uint32_t* __restrict__ fin, uint64_t n2, uint32_t newfirst) {
  uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n2) return;
  uint32_t c = cur[o];
  uint32_t f = fin[o];
  if (c < newfirst && c < f) { fin[o] = c; f = c; }
  if (f < c) cur[o] = f;
}

extern "C" void shn_ext_destroy(shn_ext* e) {
  if (!e) return;
  hipSetDevice(e->device);
  void* ptrs[] = {e->d_weight, e->d_flags, e->d_adjR, e->d_adjL, e->d_order, e->d_claim, e->d_claim2, e->d_nr, e->d_nl,
                  e->d_totw, e->d_hash};
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
  TRYE(hipMalloc(&e->d_claim, (2 * n + 1) * 4));
  TRYE(hipMalloc(&e->d_claim2, (2 * n + 1) * 4));
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
      (rc = g_shn_ws[13].get(64, &pc))) { shn_ext_destroy(e); return rc; }
  uint64_t* skeys = (uint64_t*)pk; uint32_t* svals = (uint32_t*)pv;
  unsigned long long* d_cnt = (unsigned long long*)pc;
  uint32_t* d_changed = (uint32_t*)(d_cnt + 1);
  TRYE(hipMemsetAsync(d_cnt, 0, 16, s));
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
  TRYE(hipMemcpyAsync(e->d_order, svals, ns * 4, hipMemcpyDeviceToDevice, s));
  TRYE(hipMemsetAsync(e->d_hash, 0xFF, (ns + 1) * 8, s));
  TRYE(hipMemsetAsync(e->d_claim, 0xFF, (2 * n + 1) * 4, s));
  // d_claim accumulates FINAL claims; prev/cur hold the claims of the not-yet-final walks.
  uint32_t *prev = nullptr, *cur = nullptr, *fin = e->d_claim;
  void* pprev;
  if ((rc = g_shn_ws[24].get((2 * n + 2) * 4, &pprev))) { shn_ext_destroy(e); return rc; }
  prev = (uint32_t*)pprev;
  cur = e->d_claim2;
  TRYE(hipMemsetAsync(prev, 0xFF, (2 * n + 1) * 4, s));
  int it = 0;
  uint32_t first = 0;                       // walks [0, first) are final
  bool converged = ns == 0;
  while (!converged && it < max_iterations) {
    TimerRegion t3(ctx, T_EXT_WALK);
    TRYE(hipMemsetAsync(cur, 0xFF, (2 * n + 1) * 4, s));
    TRYE(hipMemsetAsync(d_changed, 0xFF, 4, s));
    uint64_t nw = ns - first;
    hipLaunchKernelGGL(ext_walk_kernel<false>, dim3((uint32_t)cdiv(nw, EBLK)), dim3(EBLK), 0, s, e->d_order, nw, nullptr,
                       e->d_adjR, e->d_adjL, e->d_weight, prev, first, cur, e->d_nr, e->d_nl, e->d_totw, e->d_hash, d_changed,
                       t->d_keys, t->k, nullptr, nullptr);
    uint32_t ch = 0;
    TRYE(hipMemcpyAsync(&ch, d_changed, 4, hipMemcpyDeviceToHost, s));
    TRYE(hipStreamSynchronize(s));
    it++;
    if (getenv("SHN_DEBUG")) fprintf(stderr, "[shn_extend] iteration %d: first=%u lowest_changed=%u walks=%llu\n", it, first, ch, (unsigned long long)nw);
    // walk q = lowest rank that changed is final now (every lower rank was unchanged, hence final),
    // and so is every walk below it: freeze [first, q] and never recompute them.
    uint32_t newfirst = ch == UNCLAIMED ? (uint32_t)ns : ch + 1;
    hipLaunchKernelGGL(ext_freeze_kernel, dim3((uint32_t)cdiv(2 * n, 256)), dim3(256), 0, s, cur, fin, 2 * n, newfirst);
    first = newfirst;
    std::swap(prev, cur);
    if (ch == UNCLAIMED) converged = true;
  }
  e->iterations = it;
  if (!converged) { shn_ext_destroy(e); return shn_fail(SHN_ERR_INTERNAL, "shn_extend: walk fixpoint did not converge"); }
  e->d_claim2 = (cur == (uint32_t*)pprev) ? prev : cur;          // keep the malloc'd scratch (the other one is workspace)
  TRYE(hipGetLastError());
#undef TRYE
  *out = e;
  return SHN_OK;
}

extern "C" uint64_t shn_ext_n_walks(const shn_ext* e) { return e ? e->n_seeds : 0; }
extern "C" int shn_ext_iterations(const shn_ext* e) { return e ? e->iterations : 0; }

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
  HIP_TRY(hipMemsetAsync(e->d_claim2, 0xFF, (2 * e->n + 1) * 4, s));
  hipLaunchKernelGGL(ext_walk_kernel<true>, dim3((uint32_t)cdiv(n_sel, EBLK)), dim3(EBLK), 0, s, e->d_order, n_sel, d_sel,
                     e->d_adjR, e->d_adjL, e->d_weight, e->d_claim, 0u, e->d_claim2, e->d_nr, e->d_nl, e->d_totw, e->d_hash,
                     nullptr, e->table->d_keys, e->k, d_off, d_out);
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
